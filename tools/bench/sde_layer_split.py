# Where the NeuralDSDE layer's forward time goes at BASELINE config 5: the same call with pieces removed (wall time per call,
# median of 30):  python tools/bench/sde_layer_split.py
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import _mlp_desc
D, H, B = 32, 64, 512
tol = 0.14
f32 = np.float32
rng = np.random.default_rng(0)
h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
pd = (rng.standard_normal(D * H + H + H * D + D) * 0.3).astype(f32); pg = (rng.standard_normal(D * D + D) * 0.05).astype(f32)
h.set_params(pd, pg)
x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
z = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()


def path(nfine):
    hh = f32(1.0 / nfine)
    W = np.concatenate([np.zeros((1, B, D), f32), np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(hh)).astype(f32), axis=0, dtype=f32)], axis=0)
    return torch.from_numpy(W).cuda()


def med(fn, reps=30):
    ts = []
    for i in range(reps + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        if i >= 3: ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, r


for nfine in (64, 256, 1024):
    Wd = path(nfine)
    t, r = med(lambda: h.solve_adaptive(x, Wd, 0.0, 1.0, tol, tol, dt0=0.01))
    att = r["stats"]["naccept"] + r["stats"]["nreject"]
    print(f"nfine {nfine}: solve_adaptive (dt0 given)          {t:.3f} ms, {att} attempts")
    for name, kw in (("layer none, dt0 given ", dict(mode="none", dt0=0.01)), ("layer none, auto dt0  ", dict(mode="none")),
                     ("layer unbiased, dt0 given", dict(mode="unbiased", dt0=0.01)), ("layer unbiased, auto  ", dict(mode="unbiased"))):
        t, r = med(lambda: h.node_forward_record(x, Wd, 0.0, 1.0, tol, tol, z_local=z, t1_or_rand=0.4, saveat=(), save_start=-1, **kw))
        att = r["stats"]["naccept"] + r["stats"]["nreject"]
        print(f"nfine {nfine}: {name}            {t:.3f} ms, {att} attempts")
