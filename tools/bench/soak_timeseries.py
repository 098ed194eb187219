# Soak test of the time-series pullback (layer `saveat` kwarg): random saveat lists, t1, modes; overlap on vs off bit for bit;
# any solver error is reported with its parameters.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H = 784, 100
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
ps = torch.from_numpy(P.glorot_params(model, seed=2)).cuda()
h = Handle(_mlp_desc(model)); h.set_params(ps)
rng = np.random.default_rng(11)
bad = 0; err = 0; t0w = time.time()
for it in range(N):
    B = int(rng.choice([4, 64, 200])); tol = float(rng.choice([1e-3, 1e-5, 1e-7])); t1 = float(rng.random())
    mode = str(rng.choice(["unbiased", "biased", "none"]))
    ns = int(rng.integers(1, 6))
    sv = sorted(set([float(np.float32(v)) for v in rng.random(ns)] + ([1.0] if rng.random() < 0.7 else [])))
    if rng.random() < 0.2: sv = sorted(set(sv + [float(np.float32(t1))]))   # a saveat point equal to t1
    x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
    res = []
    for on in (True, False):
        try:
            h.set_overlap(on)
            fw = h.node_forward_record_ts(x, 0.0, 1.0, tol, tol, sv, mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=10000)
            du = torch.stack([torch.full_like(x, 1e-3 * (i + 1)) for i in range(fw["u"].shape[0])])
            bw = h.node_backward_recorded_ts(du, w_reg=1.5)
            res.append((fw["u"].clone(), torch.from_numpy(fw["t"]), fw["reg_val"], fw["nfe"], bw["dx"], bw["dp"]))
        except Exception as e:
            res.append((str(e),))
            if "biased mode needs at least two saved times" in str(e): continue   # the reference's rand(rng, sol.t[1:end-1]) on one saved time fails too
            err += on
            print(f"pass {it}: B={B} tol={tol} t1={t1} mode={mode} saveat={sv} overlap={on}: {e}", flush=True)
    if len(res[0]) != len(res[1]): bad += 1; print(f"MISMATCH (one order failed) pass {it}", flush=True); continue
    for a, b in zip(res[0], res[1]):
        same = torch.equal(a, b) if torch.is_tensor(a) else (a == b)
        if not same: bad += 1; print(f"MISMATCH pass {it}: B={B} tol={tol} t1={t1} mode={mode} saveat={sv}", flush=True); break
    if it % 100 == 99: print(f"{it + 1} passes, {bad} mismatches, {err} errors, {time.time() - t0w:.0f} s", flush=True)
print(f"time-series soak: {N} passes, {bad} mismatches, {err} solver errors", flush=True)
sys.exit(1 if bad or err else 0)
