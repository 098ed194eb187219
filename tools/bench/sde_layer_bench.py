# The NeuralDSDE layer at BASELINE config 5 (state 32, hidden 64, B = 512, abstol = reltol = 0.14): the recorded adaptive
# forward and the pullback through its recorded steps, wall time per call:
#   python tools/bench/sde_layer_bench.py [nfine]
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import _mlp_desc
D, H, B = 32, 64, 512
nfine = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tol = 0.14
f32 = np.float32
rng = np.random.default_rng(0)
h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
npd = D * H + H + H * D + D
pd = (rng.standard_normal(npd) * 0.3).astype(f32); pg = (rng.standard_normal(D * D + D) * 0.05).astype(f32)
h.set_params(pd, pg)
x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
hh = f32(1.0 / nfine)
W = np.concatenate([np.zeros((1, B, D), f32), np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(hh)).astype(f32), axis=0, dtype=f32)], axis=0)
Wd = torch.from_numpy(W).cuda()
z = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
du = torch.from_numpy(rng.standard_normal((1, B, D)).astype(f32)).cuda()
for mode in ("unbiased", "none"):
    tf, tb = [], []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fw = h.node_forward_record(x, Wd, 0.0, 1.0, tol, tol, z_local=z, mode=mode, t1_or_rand=0.3 + 0.03 * i, saveat=(), save_start=-1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ns = fw["u"].shape[0]
        bw = h.node_backward_recorded(du.expand(ns, B, D).contiguous(), w_reg=2.0 if mode != "none" else 0.0)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if i >= 2: tf.append(t1 - t0); tb.append(t2 - t1)
    st = fw["stats"]
    att = st["naccept"] + st["nreject"]
    print(f"{mode}: forward+record {np.median(tf)*1e3:.3f} ms ({att} attempted, {st['naccept']} accepted steps: {np.median(tf)/att*1e6:.1f} us per attempt), "
          f"pullback {np.median(tb)*1e3:.3f} ms ({np.median(tb)/st['naccept']*1e6:.1f} us per recorded step)", flush=True)
