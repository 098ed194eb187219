"""Scratch timing helper for gpurun (not a test)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H = 784, 100
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
p = P.glorot_params(model, seed=0)
for B in [int(a) for a in sys.argv[1:]] or [512]:
    h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(p))
    x = torch.rand(B, D, device="cuda")
    for tol in (1.4e-8,):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            r = h.solve(x, 0.0, 1.0, tol, tol, saveat=[1.0], maxiters=10000)
            torch.cuda.synchronize(); el = time.time() - t0
        ms, nl = h.last_solve_kernel_ms()
        st = r["stats"]
        print(f"B={B} tol={tol}: nf={st['nf']} acc={st['naccept']} rej={st['nreject']} wall={el*1e3:.2f}ms "
              f"kernel_ms={ms:.3f} launches={nl} -> {st['nf']/el:.0f} NFE/s, per-step-kernel ~{ms/nl*1e3:.1f}us")
