# prints a hash of (dx, dp, adjoint stats) of a few recorded forward + backward passes at the MNIST shape: run it under two
# settings of an environment switch and compare the lines (bitwise A/B across processes)
import hashlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, B = 784, 100, 512
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(P.glorot_params(model, seed=0)))
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
du = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32) * np.float32(1e-3)).cuda()
m = hashlib.sha256()
for t1 in (0.07, 0.43, 0.91):
    for tol in (1e-5, 1.4e-8):
        h.node_forward_record(x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="error_estimate", t1_or_rand=t1, maxiters=10000)
        b = h.node_backward_recorded(du, w_reg=2.5)
        m.update(b["dx"].cpu().numpy().tobytes()); m.update(b["dp"].cpu().numpy().tobytes())
        m.update(repr(sorted(b["stats_bwd"].items())).encode())
print(m.hexdigest())
