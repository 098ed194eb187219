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


@pytest.mark.parametrize("reg_type", ["error_estimate", "stiffness_estimate"])
@pytest.mark.parametrize("D,H,B,act,td", [(784, 100, 32, "tanh", True), (32, 64, 20, "gelu", True)])
def test_reg_gradient_matches_oracle(oracle, gpu_pkg, reg_type, D, H, B, act, td):
    """d reg_val/d ps through one local step; rtol 1e-3 on the gradient norm (fp32 reverse sweep
    through six stages), reg_val itself is bit-exact."""
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, D, H, B, act, td, scale=3.0)
    k1 = fld.rhs(x, 0.2)
    gp_ref, rv_ref = oracle.step_reg_grad(fld, x, k1, 0.2, 0.1, 1e-3, 1e-3, reg_type)
    gp, rv = h.step_reg_grad(torch.from_numpy(x).cuda(), torch.from_numpy(k1).cuda(), 0.2, 0.1, 1e-3, 1e-3, reg_type)
    assert rv == rv_ref
    assert _rel(gp.cpu().numpy(), gp_ref) < 1e-3, _rel(gp.cpu().numpy(), gp_ref)
    assert np.isfinite(gp.cpu().numpy()).all() and (gp.cpu().numpy() != 0).any()     # runtests.jl:130-131


@pytest.mark.parametrize("mode,w_reg", [("none", 0.0), ("unbiased", 0.0), ("unbiased", 2.5), ("biased", 1.0)])
def test_node_backward_matches_oracle(oracle, gpu_pkg, mode, w_reg):
    """Continuous adjoint + regulariser sweep vs the oracle: rtol 2e-4 on ||dx||, ||dp|| (both sides
    solve the adjoint ODE adaptively at abstol=reltol=1e-5; step sequences may differ)."""
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, 784, 100, 32, "tanh", True, scale=1.5)
    g = np.random.default_rng(4).standard_normal(x.shape).astype(np.float32)
    ref = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode=mode, t1_or_rand=0.43, w_reg=w_reg)
    got = h.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-5, 1e-5, torch.from_numpy(g).cuda(), mode=mode,
                          t1_or_rand=0.43, w_reg=w_reg, maxiters=10000)
    assert ref["retcode"] == 0
    assert got["stats_fwd"]["naccept"] == ref["stats_fwd"]["naccept"]       # forward is bit-exact
    dx, dp = got["dx"].cpu().numpy(), got["dp"].cpu().numpy()
    assert _rel(dx, ref["dx"]) < 2e-4, _rel(dx, ref["dx"])
    assert _rel(dp, ref["dp"]) < 2e-4, _rel(dp, ref["dp"])
    # test/runtests.jl:24-29: gradients finite and non-zero
    assert np.isfinite(dx).all() and np.isfinite(dp).all() and np.all(dx != 0) and np.mean(dp != 0) > 0.99
    print(mode, w_reg, "bwd steps gpu/oracle:", got["stats_bwd"]["naccept"], ref["stats_bwd"]["naccept"])
