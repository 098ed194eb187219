"""Pins for the oracle's backward pass against an independent float64
torch-autograd restatement: VJP of the field, gradient of the local regularisation value through
one Tsit5 step (k1, dt, uprev constant), and the continuous adjoint of the solve."""
import numpy as np
import pytest
import torch

from np_restatement import A as TA, BT as TBT

C4 = [0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0]


def _unpack(p, D, H, td):
    o = 0
    W1 = p[o:o + H * (D + td)].reshape(D + td, H).T; o += H * (D + td)
    b1 = p[o:o + H]; o += H
    W2 = p[o:o + D * (H + td)].reshape(H + td, D).T; o += D * (H + td)
    b2 = p[o:o + D]
    return W1, b1, W2, b2


def _field64(p, D, H, td, act):
    def f(u, t):
        W1, b1, W2, b2 = _unpack(p, D, H, td)
        tc = torch.full((u.shape[0], 1), float(t), dtype=torch.float64)
        x = torch.cat([u, tc], 1) if td else u
        pre = x @ W1.T + b1
        h = torch.tanh(pre) if act == "tanh" else pre * torch.sigmoid(1.5957691216057308 * pre * (1 + 0.044715 * pre * pre))
        h = torch.cat([h, tc], 1) if td else h
        return h @ W2.T + b2
    return f


def _mk(oracle, D, H, B, act, td, scale=3.0, seed=0):
    p = oracle.glorot_mlp_params(D, H, time_dep=td, seed=seed) * np.float32(scale)
    p += np.random.default_rng(seed + 1).standard_normal(p.size).astype(np.float32) * np.float32(0.05)
    x = np.random.default_rng(seed + 2).standard_normal((B, D)).astype(np.float32)
    return p, x, oracle.MlpField(D, H, p, time_dep=td, act=act, nthreads=2)


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("D,H,B,act,td", [(8, 16, 3, "tanh", True), (6, 10, 4, "gelu", True), (8, 16, 2, "tanh", False)])
def test_vjp_matches_autograd(oracle, D, H, B, act, td):
    p, x, fld = _mk(oracle, D, H, B, act, td)
    lam = np.random.default_rng(5).standard_normal((B, D)).astype(np.float32)
    dy, gp = oracle.mlp_vjp(fld, x, 0.3, lam)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    out = _field64(pt, D, H, int(td), act)(xt, 0.3)
    (out * torch.tensor(lam, dtype=torch.float64)).sum().backward()
    assert _rel(dy, xt.grad.numpy()) < 2e-6
    assert _rel(gp, pt.grad.numpy()) < 2e-6


def _step64(f, uprev, k1, t, dt, abstol, reltol):
    ks = [k1]
    xs = []
    for s in range(2, 8):
        acc = sum(a * k for a, k in zip(TA[s], ks))
        x = uprev + dt * acc
        xs.append(x)
        ks.append(f(x, t + C4[s - 2] * dt))
    u, g6 = xs[5], xs[4]
    utilde = dt * sum(b * k for b, k in zip(TBT, ks))
    r = utilde / (abstol + torch.maximum(uprev.abs(), u.abs()) * reltol)
    eest = torch.sqrt((r * r).mean())
    den = torch.sqrt(((u - g6) ** 2).mean())
    stiff = (torch.sqrt(((ks[6] - ks[5]) ** 2).mean()) / (den + 1.1920929e-7)).abs() / 3.5068
    return eest * dt, stiff


@pytest.mark.parametrize("reg_type", ["error_estimate", "stiffness_estimate"])
@pytest.mark.parametrize("D,H,B,act,td", [(8, 16, 3, "tanh", True), (6, 10, 2, "gelu", True)])
def test_reg_gradient_matches_autograd(oracle, reg_type, D, H, B, act, td):
    """d reg_val / d ps with k1, dt, uprev held constant (neural_ode.jl:40; runtests.jl:127-131)."""
    p, x, fld = _mk(oracle, D, H, B, act, td, scale=4.0)
    k1 = fld.rhs(x, 0.2)
    gp, rv = oracle.step_reg_grad(fld, x, k1, 0.2, 0.3, 1e-3, 1e-3, reg_type)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    re, rs = _step64(_field64(pt, D, H, int(td), act), torch.tensor(x, dtype=torch.float64),
                     torch.tensor(k1, dtype=torch.float64), 0.2, 0.3, 1e-3, 1e-3)
    val = re if reg_type == "error_estimate" else rs
    val.backward()
    assert np.isclose(rv, val.item(), rtol=2e-4)
    assert _rel(gp, pt.grad.numpy()) < 2e-3, _rel(gp, pt.grad.numpy())
    assert np.isfinite(gp).all() and np.any(gp != 0)


def test_adjoint_matches_discretise_then_differentiate(oracle):
    """InterpolatingAdjoint restatement vs autograd through a fine fixed-step RK4 (float64)."""
    D, H, B = 6, 12, 3
    p, x, fld = _mk(oracle, D, H, B, "tanh", True, scale=2.5)
    g = np.random.default_rng(9).standard_normal((B, D)).astype(np.float32)
    r = oracle.node_backward(fld, x, 0.0, 1.0, 1e-6, 1e-6, g, mode="unbiased", t1_or_rand=0.37, w_reg=0.0)
    assert r["retcode"] == 0 and r["stats_bwd"]["naccept"] > 0
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    f = _field64(pt, D, H, 1, "tanh")
    u, n = xt, 400
    hstep = 1.0 / n
    for i in range(n):
        t = i * hstep
        a1 = f(u, t); a2 = f(u + 0.5 * hstep * a1, t + 0.5 * hstep)
        a3 = f(u + 0.5 * hstep * a2, t + 0.5 * hstep); a4 = f(u + hstep * a3, t + hstep)
        u = u + hstep / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
    (u * torch.tensor(g, dtype=torch.float64)).sum().backward()
    assert _rel(r["dx"], xt.grad.numpy()) < 1e-4, _rel(r["dx"], xt.grad.numpy())
    assert _rel(r["dp"], pt.grad.numpy()) < 1e-4, _rel(r["dp"], pt.grad.numpy())
    # reference behaviour pins (test/runtests.jl:24-29): all finite, all non-zero
    assert np.isfinite(r["dx"]).all() and np.isfinite(r["dp"]).all()
    assert np.all(r["dx"] != 0) and np.mean(r["dp"] != 0) > 0.99


def test_backward_modes_and_reg_weight(oracle):
    D, H, B = 6, 12, 2
    p, x, fld = _mk(oracle, D, H, B, "gelu", True, scale=2.0)
    g = np.ones((B, D), np.float32)
    none = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode="none")
    unb0 = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode="unbiased", t1_or_rand=0.6, w_reg=0.0)
    unb1 = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode="unbiased", t1_or_rand=0.6, w_reg=10.0)
    bia = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode="biased", t1_or_rand=0.6, w_reg=0.0)
    for r in (none, unb0, unb1, bia):
        assert r["retcode"] == 0
    # the extra tstop at t1 changes the backward step sequence, not the gradient (to solver tolerance)
    assert _rel(unb0["dx"], none["dx"].astype(np.float64)) < 1e-3
    assert _rel(bia["dp"], none["dp"].astype(np.float64)) < 1e-3
    # reg_val has no gradient w.r.t. x (runtests.jl:129): dx is unchanged by w_reg, dp is not
    assert np.array_equal(unb0["dx"], unb1["dx"]) and not np.array_equal(unb0["dp"], unb1["dp"])


def test_classifier_ce_matches_torch_float64(oracle):
    O = oracle
    """Dense(D => K) + logitcrossentropy (experiments/src/construct.jl:199, utils.jl:88) and its cotangents"""
    rng = np.random.default_rng(3)
    B, D, K = 9, 33, 10
    u = rng.standard_normal((B, D)).astype(np.float32)
    pc = (rng.standard_normal(K * (D + 1)) * 0.3).astype(np.float32)
    lab = rng.integers(0, K, B)
    loss, lg, du, dpc = O.classifier_ce(u, pc, K, lab)
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    pt = torch.tensor(pc, dtype=torch.float64, requires_grad=True)
    W = pt[:K * D].reshape(D, K).t()  # flat = vec(W) column-major (K x D)
    logits = ut @ W.t() + pt[K * D:]
    ce = torch.nn.functional.cross_entropy(logits, torch.tensor(lab))
    ce.backward()
    assert abs(float(loss) - ce.item()) < 1e-6
    np.testing.assert_allclose(lg, logits.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(du, ut.grad.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(dpc, pt.grad.numpy(), rtol=0, atol=1e-7)
