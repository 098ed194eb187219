# per-pass cost of the per-step RCCL all-reduce on one rank (LRNDE_FORCE_COMM=1 vs unset)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, B = 784, 100, 512
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(P.glorot_params(model, seed=0)))
P.init_comm(h, 0, 1)
x = torch.from_numpy(np.random.default_rng(0).random((B, D), dtype=np.float32)).cuda()
for i in range(3): h.node_forward(x, 0.0, 1.0, 1.4e-8, 1.4e-8, mode="unbiased", t1_or_rand=0.3, maxiters=10000)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): r = h.node_forward(x, 0.0, 1.0, 1.4e-8, 1.4e-8, mode="unbiased", t1_or_rand=0.3, maxiters=10000)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
print(f"FORCE_COMM={os.environ.get('LRNDE_FORCE_COMM')}: {el*1e3:.3f} ms/pass nfe={r['nfe']} -> {r['nfe']/el:.0f} NFE/s", flush=True)
