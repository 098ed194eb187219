"""schedulers (experiments/src/utils.jl:1-68) on the host, update rules (experiments/src/construct.jl:104-126)
on the device against a numpy float64 restatement of the Optimisers.jl formulas."""
import math

import numpy as np
import pytest


def test_schedulers_follow_the_reference_formulas():
    import lrnde_amd as P
    e = P.ExponentialDecay(1e-2, 1e-4, 1000)
    assert math.isclose(e(0), 1e-2) and math.isclose(e(1000), 1e-4, rel_tol=1e-12) and math.isclose(e(500), 1e-3, rel_tol=1e-12)
    assert math.isclose(P.InverseDecay(0.1, 0.5)(4), 0.1 / 3)
    s = P.Step(0.1, 0.5, [3, 6])                      # lr0 * gamma^(searchsortedfirst(steps, t-1) - 1)
    assert [s(t) for t in (1, 4, 5, 7, 8)] == [0.1, 0.1, 0.05, 0.05, 0.025]
    assert P.Step(0.1, 0.1, 5)(7) == pytest.approx(0.01)
    c = P.CosineAnneal(0.1, 0.001, 10, restart=True, dampen=1.2)
    assert c(1) == pytest.approx(0.1) and c(11) == pytest.approx(0.1 / 1.2) and c(6) == pytest.approx((0.099 * (1 + math.cos(math.pi * 0.5)) / 2 + 0.001))
    c2 = P.CosineAnneal(0.1, 0.001, 10)
    assert c2(11) == pytest.approx(0.001)
    assert P.Constant(3e-4)(99) == 3e-4
    sch = P.construct_scheduler("exponential", 1e-3, total_steps=100, exponential_lr_div_factor=10.0)
    assert math.isclose(sch(100), 1e-4, rel_tol=1e-12)
    with pytest.raises(ValueError, match="unknown value for `scheduler`"):
        P.construct_scheduler("linear", 1e-3)
    with pytest.raises(ValueError, match="unknown value for `optimizer`"):
        P.Optimiser("rmsprop")


def _ref_update(kind, x, g, s1, s2, eta, rho, b1, b2, eps, t, wd):
    x, g = x.astype(np.float64), g.astype(np.float64)
    if kind == "sgd":
        d = eta * g
    elif kind == "momentum":
        s1[:] = rho * s1 - eta * g; d = -s1
    elif kind == "nesterov":
        d = -rho * rho * s1 + (1 + rho) * eta * g; s1[:] = rho * s1 - eta * g
    elif kind == "adam":
        s1[:] = b1 * s1 + (1 - b1) * g; s2[:] = b2 * s2 + (1 - b2) * g * g
        d = s1 / (1 - b1 ** t) / (np.sqrt(s2 / (1 - b2 ** t)) + eps) * eta
    else:
        s1[:] = b1 * s1 + (1 - b1) * g; s2[:] = np.maximum(b2 * s2, np.abs(g))
        d = eta / (1 - b1 ** t) * s1 / (s2 + eps)
    return x - (d + wd * x)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("sgd", {}), ("momentum", dict(momentum=0.9)), ("nesterov", dict(momentum=0.9, nesterov=True)),
                                     ("adam", {}), ("adam", dict(weight_decay=1e-2)), ("adamax", {})])
def test_update_rules_match_float64_restatement(gpu_pkg, name, kw):
    import torch
    P = gpu_pkg
    n = 158568
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n).astype(np.float32)
    opt = P.Optimiser("sgd" if name in ("sgd", "momentum", "nesterov") else name, learning_rate=1e-2, **kw)
    xd = torch.from_numpy(x.copy()).cuda()
    xr = x.astype(np.float64)
    s1, s2 = np.zeros(n), np.zeros(n)
    sched = P.CosineAnneal(1e-2, 1e-4, 5, restart=True)
    for t in range(1, 8):
        g = rng.standard_normal(n).astype(np.float32)
        lr = sched(t)
        opt.update(xd, torch.from_numpy(g).cuda(), lr=lr)
        xr = _ref_update(name, xr, g, s1, s2, lr, kw.get("momentum", 0.0), 0.9, 0.999, 1e-8, t, kw.get("weight_decay", 0.0))
    err = np.abs(xd.cpu().numpy() - xr).max() / np.abs(xr).max()
    assert err < 1e-6, err
