# Soak test of the batch-sharded (nranks > 1) path through the in-process local communicator on one GPU: many layer forwards
# with random t1 / tolerance / mode on 2 and 4 ranks, every rank's result compared bit for bit with the unsharded handle's.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H = 784, 100
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
p = P.glorot_params(model, seed=0)
hu = Handle(_mlp_desc(model)); hu.set_params(torch.from_numpy(p))
groups = {}
for R in (2, 4):
    lc = P.LocalComm(R); hs = []
    for r in range(R):
        h = Handle(_mlp_desc(model), stream=torch.cuda.Stream()); h.set_params(torch.from_numpy(p)); lc.join(h, r); hs.append(h)
    groups[R] = (lc, hs)
torch.cuda.synchronize()
rng = np.random.default_rng(7)
bad = 0; t0w = time.time()
for it in range(N):
    R = int(rng.choice([2, 4])); Bl = int(rng.choice([5, 64, 128])); B = R * Bl
    tol = float(rng.choice([1e-3, 1e-5, 1.4e-8])); t1 = float(rng.random()); mode = str(rng.choice(["unbiased", "biased", "none"]))
    x = rng.random((B, D), dtype=np.float32)
    xd = torch.from_numpy(x).cuda()
    ref = hu.node_forward(xd, 0.0, 1.0, tol, tol, mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=10000)
    lc, hs = groups[R]
    xs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(x, r, R))).cuda() for r in range(R)]
    torch.cuda.synchronize()
    got = P.run_ranks([(lambda r=r: hs[r].node_forward(xs[r], 0.0, 1.0, tol, tol, mode=mode, reg_type="error_estimate", t1_or_rand=t1,
                                                       maxiters=10000)) for r in range(R)])
    torch.cuda.synchronize()
    ue = ref["u_end"].cpu().numpy()
    for r in range(R):
        ok = got[r]["nfe"] == ref["nfe"] and got[r]["reg_val"] == ref["reg_val"] and got[r]["stats"] == ref["stats"] and \
            np.array_equal(got[r]["u_end"].cpu().numpy(), P.shard_columns(ue, r, R))
        if not ok:
            bad += 1; print(f"MISMATCH pass {it}: R={R} Bl={Bl} tol={tol} t1={t1} mode={mode} rank {r}", flush=True); break
    if it % 50 == 49: print(f"{it + 1} passes, {bad} mismatches, {time.time() - t0w:.0f} s", flush=True)
print(f"sharded soak: {N} passes, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
