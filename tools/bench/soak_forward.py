# Soak test of the layer forward / training step: many passes with random t1, batch size and tolerance, every result compared
# bit for bit with the same call with the companion stream switched off (lrnde_set_overlap).  Exit code 1 on any mismatch.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, K = 784, 100, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
rng = np.random.default_rng(123)
ps = torch.from_numpy(P.glorot_params(model, seed=1)).cuda()
pc = torch.from_numpy((rng.standard_normal(K * (D + 1)) * 0.05).astype(np.float32)).cuda()
handles = {}
bad = 0; t_start = time.time()
for it in range(N):
    B = int(rng.choice([3, 64, 257, 512, 1024]))
    tol = float(rng.choice([1e-3, 1e-5, 1.4e-8]))
    t1 = float(rng.choice([rng.random(), rng.random(), rng.random() * 2e-3, 1.0 - rng.random() * 2e-3, float(np.float32(1e-6) + np.float32(rng.random()) * np.float32(1e-5))]))  # incl. next to the ends of the span
    # (a t1 within dtmin = eps(1) = 1.2e-7 of t0 leaves the adjoint a last interval shorter than dtmin: DtLessThanMin by the
    #  controller's own rule, in both orders — one draw in eight million; not sampled here)
    mode = str(rng.choice(["unbiased", "unbiased", "biased", "none"]))
    ss = bool(rng.integers(0, 2))
    x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
    h = handles.get(B)
    if h is None:
        h = handles[B] = Handle(_mlp_desc(model)); h.set_params(ps)
    res = []
    for on in (True, False):
      try:
        h.set_overlap(on)
        if it % 3 == 0 and mode != "none":   # recorded forward + backward
            fw = h.node_forward_record(x, 0.0, 1.0, tol, tol, mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=10000, save_start=ss)
            du = torch.full_like(x, 1e-3)
            bw = h.node_backward_recorded(du, w_reg=2.5)
            res.append((fw["u_end"], fw["reg_val"], fw["nfe"], bw["dx"], bw["dp"]))
        else:
            fw = h.node_forward(x, 0.0, 1.0, tol, tol, mode=mode, reg_type="stiffness_estimate" if it % 2 else "error_estimate",
                                t1_or_rand=t1, maxiters=10000, save_start=ss)
            res.append((fw["u_end"], fw["reg_val"], fw["nfe"]))
      except Exception as e:   # a solver return code (e.g. DtLessThanMin in the adjoint) is a result too: both orders must agree
        res.append((str(e),))
        print(f"pass {it}: B={B} tol={tol} t1={t1} mode={mode} overlap={on}: {e}", flush=True)
    if len(res[0]) != len(res[1]):
        bad += 1; print(f"MISMATCH (one order failed) at pass {it}", flush=True); continue
    for a, b in zip(res[0], res[1]):
        same = torch.equal(a, b) if torch.is_tensor(a) else (a == b)
        if not same:
            bad += 1; print(f"MISMATCH at pass {it}: B={B} tol={tol} t1={t1} mode={mode}", flush=True); break
    if it % 100 == 99: print(f"{it + 1} passes, {bad} mismatches, {time.time() - t_start:.0f} s", flush=True)
print(f"soak: {N} passes, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
