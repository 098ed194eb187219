# Soak test of the adaptive Euler-Heun solve (device-controlled on the 32/64 shape, host-controlled elsewhere): random batch,
# tolerance, grid, first step and drift scale; every solve must end at t1 with finite values, and the trace must be consistent
# (accepted steps add up to the span, rejected ones do not move).
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import _mlp_desc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
f32 = np.float32
rng = np.random.default_rng(5)
hs = {}
bad = 0; t0w = time.time()
for it in range(N):
    D, H = [(32, 64), (32, 64), (16, 24)][int(rng.integers(0, 3))]
    B = int(rng.choice([1, 9, 64, 512])); nfine = int(rng.choice([16, 64, 256])); tol = float(rng.choice([0.2, 0.05, 0.01]))
    scale = float(rng.choice([0.2, 1.0, 3.0]))
    key = (D, H)
    if key not in hs: hs[key] = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
    hd = hs[key]
    pd = (rng.standard_normal(D * H + H + H * D + D) * 0.3 * scale).astype(f32); pg = (rng.standard_normal(D * D + D) * 0.05).astype(f32)
    hd.set_params(pd, pg)
    x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
    h = f32(1.0 / nfine)
    W = np.concatenate([np.zeros((1, B, D), f32), np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(h)).astype(f32), axis=0, dtype=f32)], axis=0)
    try:
        r = hd.solve_adaptive(x, torch.from_numpy(W).cuda(), 0.0, 1.0, tol, tol, dt0=float(h) * int(rng.choice([1, 4, 16])), maxiters=20000)
        tr = r["trace"]; st = r["stats"]
        acc = tr[tr["accepted"] != 0]
        ok = bool(torch.isfinite(r["u_end"]).all()) and abs(float(acc["dt"].sum()) - 1.0) < 1e-4 and st["naccept"] == len(acc) and \
            st["nreject"] == len(tr) - len(acc) and abs(st["t_final"] - 1.0) < 1e-6
        if not ok: bad += 1; print(f"INCONSISTENT pass {it}: D={D} B={B} nfine={nfine} tol={tol} scale={scale}: {st}", flush=True)
    except Exception as e:
        if "DtLessThanMin" in str(e) and scale == 3.0: continue   # a drift too rough for the path's grid: the documented outcome
        bad += 1; print(f"pass {it}: D={D} B={B} nfine={nfine} tol={tol} scale={scale}: {e}", flush=True)
    if it % 100 == 99: print(f"{it + 1} passes, {bad} problems, {time.time() - t0w:.0f} s", flush=True)
print(f"SDE soak: {N} passes, {bad} problems", flush=True)
sys.exit(1 if bad else 0)
