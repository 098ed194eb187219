"""the layer's `saveat` time series and its pullback (src/utils.jl:25-46, src/layers/neural_ode.jl:107-111,
consumer experiments/src/construct.jl:244-249).  Cotangents on every saved state enter the reversed-time adjoint solve as
impulses on lambda at their times.  Pinned against float64 torch autograd through a fine fixed-step RK4 integration of the
same field (discretise-then-differentiate at an accuracy far beyond the adaptive solve's tolerance): no code shared with
the library or the oracle.  Tolerance 3e-4 of each gradient's norm (adaptive forward and adjoint at 1e-6, fp32)."""
import numpy as np
import pytest
import torch

from test_oracle_backward import _field64

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _reference_grads(p, x, D, H, times, cots, nsteps=200):
    """d/d(x, p) of sum_i <cots[i], u(times[i])> by float64 autograd through RK4 with nsteps steps on (0, 1)"""
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    f = _field64(pt, D, H, 1, "tanh")
    h = 1.0 / nsteps
    u, loss = xt, 0.0
    marks = {int(round(t * nsteps)): i for i, t in enumerate(times)}
    assert all(abs(k / nsteps - times[i]) < 1e-12 for k, i in marks.items())
    if 0 in marks:
        loss = loss + (u * torch.tensor(cots[marks[0]], dtype=torch.float64)).sum()
    for k in range(nsteps):
        t = k * h
        k1 = f(u, t); k2 = f(u + 0.5 * h * k1, t + 0.5 * h); k3 = f(u + 0.5 * h * k2, t + 0.5 * h); k4 = f(u + h * k3, t + h)
        u = u + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        if k + 1 in marks:
            loss = loss + (u * torch.tensor(cots[marks[k + 1]], dtype=torch.float64)).sum()
    loss.backward()
    return xt.grad.numpy(), pt.grad.numpy()


def _mk(pkg, D, H, B, scale=2.0, seed=0):
    model = pkg.TDChain(pkg.Chain(pkg.Dense(D + 1, H, "tanh"), pkg.Dense(H + 1, D)))
    p = pkg.glorot_params(model, seed=seed) * np.float32(scale)
    p = (p + np.random.default_rng(seed + 1).standard_normal(p.size).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    x = np.random.default_rng(seed + 2).random((B, D), dtype=np.float32)
    return model, p, x


@pytest.mark.parametrize("D,H,B,regularize,save_start", [(784, 100, 16, "none", False), (32, 64, 12, "unbiased", False),
                                                          (32, 64, 12, "biased", True), (2, 4, 5, "unbiased", False)])
def test_timeseries_pullback_matches_float64_autograd(gpu_pkg, D, H, B, regularize, save_start):
    """the shapes cover the device-controlled adjoint loop (4-column kernels) and the host-controlled one (D = 2)"""
    P = gpu_pkg
    model, p, x = _mk(P, D, H, B)
    times = [0.25, 0.5, 1.0]
    node = P.NeuralODE(model, regularize=regularize, abstol=1e-6, reltol=1e-6, saveat=times, save_start=save_start, maxiters=10000)
    st = node.initialstates(np.random.default_rng(3))
    xd, ps = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    sol, st2 = node(xd, ps, st)
    want_t = ([0.0] if save_start else []) + times
    assert [float(t) for t in sol.t] == want_t
    ts = P.diffeqsol_to_timeseries(sol)                       # (nseries, B, D): the round trip of src/utils.jl:42-46
    assert ts.shape == (len(want_t), B, D)
    rng = np.random.default_rng(11)
    cots = rng.standard_normal((len(want_t), B, D)).astype(np.float32)
    dx, dp, info = node.pullback(xd, ps, st, torch.from_numpy(cots).cuda(), w_reg=0.0)
    assert torch.equal(info["sol_u"], ts) and [float(t) for t in info["sol_t"]] == want_t   # pullback's forward == __call__'s
    gx, gp = _reference_grads(p, x, D, H, want_t, cots)
    assert _rel(dx.cpu().numpy(), gx) < 3e-4, _rel(dx.cpu().numpy(), gx)
    assert _rel(dp.cpu().numpy(), gp) < 3e-4, _rel(dp.cpu().numpy(), gp)
    print(f"D={D} {regularize}: dx rel {_rel(dx.cpu().numpy(), gx):.2e} dp rel {_rel(dp.cpu().numpy(), gp):.2e}, "
          f"adjoint steps {info['stats_bwd']['naccept']}")
    # a cotangent on the last state only == the plain end-state pullback of the same layer
    dxe, dpe, _ = node.pullback(xd, ps, st, torch.from_numpy(cots[-1]).cuda(), w_reg=0.0)
    node_end = P.NeuralODE(model, regularize=regularize, abstol=1e-6, reltol=1e-6, save_start=save_start, maxiters=10000)
    dxp, dpp, _ = node_end.pullback(xd, ps, st, torch.from_numpy(cots[-1]).cuda(), w_reg=0.0)
    assert _rel(dxe.cpu().numpy(), dxp.cpu().numpy()) < 2e-4 and _rel(dpe.cpu().numpy(), dpp.cpu().numpy()) < 2e-4
    # with the regulariser: the extra term is w_reg * d reg_val / d ps, no gradient to x (test/runtests.jl:127-131)
    if regularize != "none":
        dxr, dpr, infr = node.pullback(xd, ps, st, torch.from_numpy(cots).cuda(), w_reg=3.0)
        assert infr["reg_val"] == st2["reg_val"] and infr["reg_val"] > 0
        assert _rel(dxr.cpu().numpy(), dx.cpu().numpy()) < 1e-5
        assert not torch.equal(dpr, dp)


def test_timeseries_pullback_rejects_wrong_series_length(gpu_pkg):
    P = gpu_pkg
    model, p, x = _mk(P, 32, 64, 4)
    node = P.NeuralODE(model, regularize="none", abstol=1e-4, reltol=1e-4, saveat=[0.5, 1.0], save_start=False)
    st = node.initialstates(np.random.default_rng(0))
    xd, ps = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    with pytest.raises(ValueError, match="cotangents"):
        node.pullback(xd, ps, st, torch.zeros((3, 4, 32), device="cuda"))


import os as _os


@pytest.mark.parametrize("seed", list(range(int(_os.environ.get("LRNDE_SOAK_SEEDS", "8")))))
def test_timeseries_forward_soak_against_the_oracle_solve(oracle, gpu_pkg, seed):
    """`lrnde_node_forward_record_ts` with random user saveat (duplicates of t1, points at the ends), mode, t1, save_start: the
    returned series is the oracle's plain solve on the user's saveat bit for bit (saveat points are interpolated, the extra t1
    changes no step), with the reference's filter `t1 .!= sol.t` applied (src/utils.jl:25-33), and reg_val / nfe are the
    oracle's layer forward's with that t1.  LRNDE_SOAK_SEEDS=N runs N seeds."""
    import torch
    from test_gpu_parity import _mk, _eq
    rng = np.random.default_rng(40_000 + seed)
    D, H = [(784, 100), (32, 64), (20, 40), (452, 112)][int(rng.integers(0, 4))]
    B = int(rng.choice([1, 4, 9, 33])); tol = float(rng.choice([1e-3, 1e-5, 1e-7]))
    mode = str(rng.choice(["unbiased", "unbiased", "none"]))
    t1 = float(np.float32(rng.random()))
    save_start = bool(rng.integers(0, 2))
    ns = int(rng.integers(1, 7))
    sv = sorted(set(float(np.float32(v)) for v in rng.random(ns)) | ({1.0} if rng.random() < 0.7 else set()))
    if rng.random() < 0.25: sv = sorted(set(sv) | {t1})
    fld, h, p, x, _ = _mk(oracle, gpu_pkg, D, H, B, "tanh", True, scale=1.5, seed=seed)
    xd = torch.from_numpy(x).cuda()
    what = f"seed={seed} D={D} B={B} tol={tol} {mode} t1={t1} saveat={sv} save_start={save_start}"
    fw = h.node_forward_record_ts(xd, 0.0, 1.0, tol, tol, sv, mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=20000,
                                  save_start=save_start)
    ref = oracle.solve(fld, x, 0.0, 1.0, tol, tol, saveat=sv, maxiters=20000, save_start=save_start, save_everystep=False)
    keep = [i for i, tv in enumerate(ref["t"]) if not (mode == "unbiased" and tv == np.float32(t1))]
    _eq(fw["t"], ref["t"][keep], "sol.t " + what)
    _eq(fw["u"].cpu().numpy(), ref["u"][keep], "sol.u " + what)
    lay = oracle.node_forward(fld, x, 0.0, 1.0, tol, tol, mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=20000,
                              save_start=save_start)
    assert fw["reg_val"] == lay["reg_val"] and fw["nfe"] == lay["nfe"], what


@pytest.mark.parametrize("seed", list(range(int(_os.environ.get("LRNDE_SOAK_SEEDS", "6")))))
def test_timeseries_pullback_soak_against_float64_autograd(gpu_pkg, seed):
    """random saved times (on the reference integration's grid), number of series points, shape, mode and save_start: the
    pullback of one random cotangent per saved state against float64 autograd through RK4 (3e-4 of each gradient's norm)"""
    P = gpu_pkg
    rng = np.random.default_rng(60_000 + seed)
    D, H = [(32, 64), (8, 16), (2, 4), (20, 40)][int(rng.integers(0, 4))]
    B = int(rng.choice([1, 3, 8])); regularize = str(rng.choice(["none", "unbiased", "biased"])); save_start = bool(rng.integers(0, 2))
    nt = int(rng.integers(1, 6))
    ks = sorted(set(int(v) for v in rng.integers(1, 200, nt)) | ({200} if rng.random() < 0.6 else set()))
    if regularize == "biased" and len(ks) + int(save_start) < 2: ks = sorted(set(ks) | {200, 77})
    times = [k / 200.0 for k in ks]
    model, p, x = _mk(P, D, H, B, seed=seed)
    node = P.NeuralODE(model, regularize=regularize, abstol=1e-6, reltol=1e-6, saveat=times, save_start=save_start, maxiters=10000)
    st = node.initialstates(np.random.default_rng(seed))
    xd, ps = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    sol, _ = node(xd, ps, st)
    got_t = [float(t) for t in sol.t]
    want_t = ([0.0] if save_start else []) + [float(np.float32(t)) for t in times]
    # (:unbiased drops a user point that coincides with the drawn t1, src/utils.jl:31-33: cannot happen on this grid)
    assert got_t == want_t, (got_t, want_t)
    cots = rng.standard_normal((len(want_t), B, D)).astype(np.float32)
    dx, dp, info = node.pullback(xd, ps, st, torch.from_numpy(cots).cuda(), w_reg=0.0)
    gx, gp = _reference_grads(p, x, D, H, ([0.0] if save_start else []) + times, cots)
    what = f"seed={seed} D={D} B={B} {regularize} save_start={save_start} times={times}"
    assert _rel(dx.cpu().numpy(), gx) < 3e-4 and _rel(dp.cpu().numpy(), gp) < 3e-4, (what, _rel(dx.cpu().numpy(), gx), _rel(dp.cpu().numpy(), gp))
