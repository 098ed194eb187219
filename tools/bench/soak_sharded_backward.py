# Soak test of the batch-sharded backward pass (host-controlled sharded adjoint) through the in-process local communicator:
# random t1 (incl. next to the ends) / tolerance on 2 and 3 ranks vs the unsharded handle (2e-4 of the norm, replicated dp).
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H = 784, 100
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
p = P.glorot_params(model, seed=0)
hu = Handle(_mlp_desc(model)); hu.set_params(torch.from_numpy(p))
groups = {}
for R in (2, 3):
    lc = P.LocalComm(R); hs = []
    for r in range(R):
        h = Handle(_mlp_desc(model), stream=torch.cuda.Stream()); h.set_params(torch.from_numpy(p)); lc.join(h, r); hs.append(h)
    groups[R] = (lc, hs)
torch.cuda.synchronize()
rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
rng = np.random.default_rng(17)
bad = 0; t0w = time.time()
for it in range(N):
    R = int(rng.choice([2, 3])); Bl = int(rng.choice([8, 32])); B = R * Bl
    tol = float(rng.choice([1e-3, 1e-5])); mode = str(rng.choice(["unbiased", "biased", "none"]))
    t1 = float(rng.choice([rng.random(), rng.random() * 2e-3 + 1e-5, 1.0 - rng.random() * 2e-3]))
    x = rng.random((B, D), dtype=np.float32); g = (rng.standard_normal((B, D)) * 1e-2).astype(np.float32)
    try:
        ref = hu.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, torch.from_numpy(g).cuda(), mode=mode, t1_or_rand=t1,
                               w_reg=2.5, maxiters=5000)
        lc, hs = groups[R]
        xs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(x, r, R))).cuda() for r in range(R)]
        gs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(g, r, R))).cuda() for r in range(R)]
        torch.cuda.synchronize()
        got = P.run_ranks([(lambda r=r: hs[r].node_backward(xs[r], 0.0, 1.0, tol, tol, gs[r], mode=mode, t1_or_rand=t1, w_reg=2.5,
                                                            maxiters=5000)) for r in range(R)])
        torch.cuda.synchronize()
        dx = np.concatenate([q["dx"].cpu().numpy() for q in got], axis=0)
        e1 = rel(dx, ref["dx"].cpu().numpy()); e2 = max(rel(q["dp"].cpu().numpy(), ref["dp"].cpu().numpy()) for q in got)
        repl = all(np.array_equal(q["dp"].cpu().numpy(), got[0]["dp"].cpu().numpy()) for q in got)
        if not (e1 < 5e-4 and e2 < 5e-4 and repl):
            bad += 1; print(f"MISMATCH pass {it}: R={R} Bl={Bl} tol={tol} t1={t1} mode={mode}: dx {e1:.2e} dp {e2:.2e} replicated {repl}", flush=True)
    except Exception as e:
        bad += 1; print(f"pass {it}: R={R} Bl={Bl} tol={tol} t1={t1} mode={mode}: {e}", flush=True)
    if it % 25 == 24: print(f"{it + 1} passes, {bad} problems, {time.time() - t0w:.0f} s", flush=True)
print(f"sharded backward soak: {N} passes, {bad} problems", flush=True)
sys.exit(1 if bad else 0)
