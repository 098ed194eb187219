# adaptive Euler-Heun solve of the MNIST-SDE shape (state 32, hidden 64, B=512) on a caller-supplied Brownian path:
# device-controlled loop (default) vs LRNDE_SDE_HOST_LOOP=1 (one stream sync per attempted step)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import _mlp_desc
D, H, B, nfine = 32, 64, 512, 256
f32 = np.float32
rng = np.random.default_rng(0)
hd = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
npd = D * H + H + H * D + D
pd = (rng.standard_normal(npd) * 0.3).astype(f32); pg = (rng.standard_normal(D * D + D) * 0.05).astype(f32)
hd.set_params(pd, pg)
x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
h = f32(1.0 / nfine)
W = np.concatenate([np.zeros((1, B, D), f32), np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(h)).astype(f32), axis=0, dtype=f32)], axis=0)
Wd = torch.from_numpy(W).cuda()
for _ in range(3): r = hd.solve_adaptive(x, Wd, 0.0, 1.0, 0.02, 0.02, dt0=4 * float(h))
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N): r = hd.solve_adaptive(x, Wd, 0.0, 1.0, 0.02, 0.02, dt0=4 * float(h))
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / N
att = r["stats"]["naccept"] + r["stats"]["nreject"]
print(f"HOST_LOOP={os.environ.get('LRNDE_SDE_HOST_LOOP')}: {el*1e3:.3f} ms per solve, {att} attempted steps ({r['stats']['naccept']} accepted) -> {el/att*1e6:.1f} us per attempted step", flush=True)
