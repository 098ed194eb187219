"""GPU parity of the backward pass against the oracle.  The bar is a tolerance (written below):
the device kernels sum in different (fixed) orders than the oracle's loops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)


def _mk(O, pkg, D, H, B, act, td, scale=2.0, seed=0):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    chain = pkg.Chain(pkg.Dense(D + int(td), H, act), pkg.Dense(H + int(td), D))
    model = pkg.TDChain(chain) if td else chain
    p = pkg.glorot_params(model, seed=seed) * np.float32(scale)
    p = p + np.random.default_rng(seed + 1).standard_normal(p.size).astype(np.float32) * np.float32(0.02)
    x = np.random.default_rng(seed + 2).random((B, D), dtype=np.float32)
    fld = O.MlpField(D, H, p, time_dep=td, act=act, nthreads=8)
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(p))
    return fld, h, p, x


@pytest.mark.parametrize("D,H,B,act,td", [(784, 100, 64, "tanh", True), (784, 100, 37, "tanh", True),
                                          (32, 64, 33, "gelu", True), (20, 40, 17, "tanh", False), (2, 4, 3, "gelu", True)])
def test_vjp_matches_oracle(oracle, gpu_pkg, D, H, B, act, td):
    """rtol 2e-5 on the L2 norm of each output (fp32, different summation orders)."""
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, D, H, B, act, td)
    lam = np.random.default_rng(9).standard_normal((B, D)).astype(np.float32)
    dy_ref, gp_ref = oracle.mlp_vjp(fld, x, 0.3, lam)
    dy, gp = h.vjp(torch.from_numpy(x).cuda(), 0.3, torch.from_numpy(lam).cuda())
    assert _rel(dy.cpu().numpy(), dy_ref) < 2e-5
    assert _rel(gp.cpu().numpy(), gp_ref) < 2e-5
