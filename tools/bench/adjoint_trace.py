"""Per-attempt trace of the adjoint solve: oracle vs the device-controlled loop vs the host-controlled loop (LRNDE_ADJ_HOST=1,
run this script once with and once without it).  VERDICT r2 item 1: find the first attempt where they part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import oracle as O
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc

D, H, B = 784, 100, int(os.environ.get("B", 32))
chain = P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D))
model = P.TDChain(chain)
p = P.glorot_params(model, seed=0) * np.float32(1.5)
p = p + np.random.default_rng(1).standard_normal(p.size).astype(np.float32) * np.float32(0.02)
x = np.random.default_rng(2).random((B, D), dtype=np.float32)
fld = O.MlpField(D, H, p, time_dep=True, act="tanh", nthreads=8)
g = np.random.default_rng(4).standard_normal(x.shape).astype(np.float32)
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(p))
for mode, w in [("none", 0.0), ("unbiased", 0.0), ("biased", 1.0)]:
    ref = O.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode=mode, t1_or_rand=0.43, w_reg=w, trace=True)
    h.set_adjoint_trace(4096)
    got = h.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-5, 1e-5, torch.from_numpy(g).cuda(), mode=mode,
                          t1_or_rand=0.43, w_reg=w, maxiters=10000)
    tr = h.adjoint_trace()
    print(mode, "oracle", ref["stats_bwd"]["naccept"], ref["stats_bwd"]["nreject"], "gpu", got["stats_bwd"]["naccept"],
          got["stats_bwd"]["nreject"], "dt_init", ref["stats_bwd"]["dt_init"], got["stats_bwd"]["dt_init"])
    rt = ref["trace_bwd"]
    for i in range(max(len(rt), len(tr))):
        a = "s=%.7f dt=%.6e eest=%.5e acc=%d" % rt[i] if i < len(rt) else " " * 50
        b = "s=%.7f dt=%.6e eest=%.5e acc=%d" % tr[i] if i < len(tr) else ""
        print("  %2d  O %s | G %s" % (i, a, b))
