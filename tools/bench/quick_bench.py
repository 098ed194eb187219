"""Scratch timing helper for gpurun (not a test)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    k1 = h.rhs(x, 0.0)
    # rhs timing
    for _ in range(5): h.rhs(x, 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): h.rhs(x, 0.1)
    e1.record(); torch.cuda.synchronize()
    rhs_us = e0.elapsed_time(e1) * 10
    us = h.bench_step(x, k1, 0.0, 0.02, 1.4e-8, 1.4e-8, reps=100)
    tol = 1.4e-8
    h.last_solve_kernel_ms()  # arm the solve's event pair
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        r = h.solve(x, 0.0, 1.0, tol, tol, saveat=[1.0], maxiters=10000)
        torch.cuda.synchronize(); el = time.time() - t0
    ms, nl = h.last_solve_kernel_ms()
    st = r["stats"]
    fl = 6 * 315368 * B
    print(f"B={B}: rhs {rhs_us:.1f}us  step {us:.1f}us ({fl/us/1e6:.1f} TFLOP/s, {fl/us/1e6/157.3*100:.1f}% of f32 MFMA peak) | "
          f"solve nf={st['nf']} acc={st['naccept']} rej={st['nreject']} wall={el*1e3:.2f}ms launches={nl} -> {st['nf']/el:.0f} NFE/s")
