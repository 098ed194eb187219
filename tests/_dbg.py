import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import lrnde_amd as P, oracle as O
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H = int(sys.argv[1]), int(sys.argv[2])
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
p = P.glorot_params(model, seed=0)
fld = O.MlpField(D, H, p, nthreads=8)
for B in (16, 48):
    x = np.random.default_rng(1).random((B, D), dtype=np.float32)
    k1 = fld.rhs(x, 0.1)
    ref = O.tsit5_step(fld, x, k1, 0.1, 0.05, 1e-4, 1e-4)
    h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(p))
    got = h.perform_step(torch.from_numpy(x).cuda(), torch.from_numpy(k1).cuda(), 0.1, 0.05, 1e-4, 1e-4)
    for key in ("u", "k7"):
        bad = (got[key].cpu().numpy() != ref[key])
        print(B, key, "bad samples:", np.nonzero(bad.any(axis=1))[0].tolist()[:40], "bad cols in first bad sample:",
              (np.nonzero(bad[np.nonzero(bad.any(axis=1))[0][0]])[0][:10].tolist() if bad.any() else None))
    print(B, "eest", got["eest"], ref["eest"], "stiff", got["reg_stiff"], ref["reg_stiff"])
