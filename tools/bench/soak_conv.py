# Soak test of the conv field's layer forward + backward (small images): random t1 (incl. values next to the ends of the span),
# tolerance and mode; any solver error is reported with its parameters; results must be finite.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = Hh = 8
h = P.ConvHandle(W, Hh, 8, 64, act="gelu", bn_train=True)
h.set_params(torch.from_numpy(P.glorot_conv_params(8, 64, seed=0)))
rng = np.random.default_rng(3)
err = 0; t0w = time.time()
for it in range(N):
    B = int(rng.choice([2, 4])); tol = float(rng.choice([1e-2, 1e-3, 1e-4]))
    t1 = float(rng.choice([rng.random(), rng.random() * 3e-3, 1.0 - rng.random() * 3e-3]))
    mode = str(rng.choice(["unbiased", "biased", "none"]))
    x = torch.from_numpy(rng.standard_normal((B, 8, Hh, W)).astype(np.float32)).cuda()
    du = torch.from_numpy((rng.standard_normal((B, 8, Hh, W)) * 1e-2).astype(np.float32)).cuda()
    try:
        r = h.node_backward(x, 0.0, 1.0, tol, tol, du, mode=mode, reg_type="error_estimate", t1_or_rand=t1, w_reg=1.0, maxiters=5000)
        ok = bool(torch.isfinite(r["dx"]).all()) and bool(torch.isfinite(r["dp"]).all())
        if not ok: err += 1; print(f"pass {it}: non-finite gradients B={B} tol={tol} t1={t1} mode={mode}", flush=True)
    except Exception as e:
        err += 1; print(f"pass {it}: B={B} tol={tol} t1={t1} mode={mode}: {e}", flush=True)
    if it % 50 == 49: print(f"{it + 1} passes, {err} errors, {time.time() - t0w:.0f} s", flush=True)
print(f"conv soak: {N} passes, {err} errors", flush=True)
sys.exit(1 if err else 0)
