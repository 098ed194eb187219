"""VCAB3 / VCABM3 as the layer's global solver (experiments/src/construct.jl:154-164): the HIP path (csrc/lrnde_adams.hpp,
through lrnde_set_solver + the ordinary entry points) against the oracle's restatement of the same published algorithm
(oracle/lrnde_oracle.c adams_solve) — bit for bit, both are written in one arithmetic — and against an independent
float64 solution.  UPSTREAM-RECALL: OrdinaryDiffEq's source for these methods is not in the reference; what is pinned here
is the product against the restatement, and the restatement against the mathematics (tests/test_oracle_adams.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _trace_equal(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    for i, (a, b) in enumerate(zip(got, ref)):
        assert tuple(a) == tuple(b), (i, tuple(a), tuple(b))


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
@pytest.mark.parametrize("D,H,B,act,td,tol", [(784, 100, 64, "tanh", True, 1e-4), (32, 64, 33, "gelu", True, 1e-5),
                                              (20, 40, 17, "tanh", False, 1e-3), (2, 4, 3, "gelu", True, 1e-6)])
def test_adams_solve_equals_the_oracle_attempt_by_attempt(oracle, gpu_pkg, solver, D, H, B, act, td, tol):
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, D, H, B, act, td)
    h.set_solver(solver)
    sv = [0.0, 0.31, 0.31, 0.77, 1.0]
    ref = oracle.solve(fld, x, 0.0, 1.0, tol, tol, saveat=sv, save_start=True, solver=solver)
    got = h.solve(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, saveat=sv, save_start=True, trace=True)
    assert ref["retcode"] == 0 and got["retcode"] == 0
    _trace_equal(got["trace"], ref["trace"])
    for k in ("nf", "naccept", "nreject", "iters", "nsaved", "t_final", "eest_last", "dt_init"):
        assert got["stats"][k] == ref["stats"][k], (k, got["stats"], ref["stats"])
    assert np.array_equal(got["t"], ref["t"])
    assert np.array_equal(got["u"].cpu().numpy(), ref["u"])
    # the method's own bookkeeping: 3 evaluations for the start (initdt 2 + fsalfirst), 3 per Bogacki-Shampine start-up step
    # (two accepted ones, and every rejected attempt before the third acceptance), then 1 (VCAB3) or 2 (VCABM3) per attempt
    per = 2 if solver == "vcabm3" else 1
    tr = ref["trace"]
    nstart = 0
    acc = 0
    for r in tr:
        if acc < 2:
            nstart += 1
        acc += int(r["accepted"])
    assert ref["stats"]["nf"] == 3 + 3 * nstart + per * (len(tr) - nstart)


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_adams_every_step_series_equals_the_oracle(oracle, gpu_pkg, solver):
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, 32, 64, 16, "tanh", True)
    h.set_solver(solver)
    ref = oracle.solve(fld, x, 0.0, 1.0, 1e-4, 1e-4, solver=solver)
    got = h.solve(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-4, 1e-4)
    assert np.array_equal(got["t"], ref["t"]) and np.array_equal(got["u"].cpu().numpy(), ref["u"])
    assert len(ref["t"]) == ref["stats"]["naccept"] > 10


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_adams_solution_against_float64(oracle, gpu_pkg, solver):
    """Independent of either restatement: the same field integrated in float64 by scipy's DOP853 at 1e-12.  The Adams
    solves converge to it as their tolerance shrinks (global error of an order-3 method at tolerance tol ~ tol^(3/4)..tol)."""
    import torch
    from scipy.integrate import solve_ivp
    D, H, B = 12, 24, 5
    fld, h, p, x = _mk(oracle, gpu_pkg, D, H, B, "tanh", True, scale=3.0)
    W1 = p[:H * (D + 1)].reshape(D + 1, H).T.astype(np.float64); b1 = p[H * (D + 1):H * (D + 1) + H].astype(np.float64)
    o2 = H * (D + 1) + H
    W2 = p[o2:o2 + D * (H + 1)].reshape(H + 1, D).T.astype(np.float64); b2 = p[o2 + D * (H + 1):].astype(np.float64)

    def f(t, y):
        y = y.reshape(B, D)
        hh = np.tanh(np.concatenate([y, np.full((B, 1), t)], 1) @ W1.T + b1)
        return (np.concatenate([hh, np.full((B, 1), t)], 1) @ W2.T + b2).reshape(-1)
    ex = solve_ivp(f, (0.0, 1.0), x.astype(np.float64).reshape(-1), method="DOP853", rtol=1e-12, atol=1e-12).y[:, -1].reshape(B, D)
    h.set_solver(solver)
    errs = []
    for tol in (1e-3, 1e-4, 1e-5):
        got = h.solve(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, saveat=[1.0])
        errs.append(np.abs(got["u"][-1].cpu().numpy() - ex).max() / np.abs(ex).max())
    print(solver, "rel err vs float64 at tol 1e-3, 1e-4, 1e-5:", ["%.2e" % e for e in errs])
    assert errs[0] < 5e-2 and errs[1] < 1e-2 and errs[2] < 2e-3 and errs[2] < errs[0]


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
@pytest.mark.parametrize("mode", ["none", "unbiased", "biased"])
def test_layer_forward_with_an_adams_solver_equals_the_oracle(oracle, gpu_pkg, solver, mode):
    """neural_ode.jl:56-100 with n.solver = VCAB3() / VCABM3(): the global solve by the Adams method, sol(t1) from its
    Hermite interpolant, the local regularisation step by Tsit5 (the reference builds that integrator with Tsit5())."""
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, 784, 100, 32, "tanh", True, scale=1.5)
    h.set_solver(solver)
    ref = oracle.node_forward(fld, x, 0.0, 1.0, 1e-4, 1e-4, mode=mode, t1_or_rand=0.37, solver=solver)
    got = h.node_forward(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-4, 1e-4, mode=mode, t1_or_rand=0.37)
    assert np.array_equal(got["u_end"].cpu().numpy(), ref["u_end"])
    assert got["nfe"] == ref["nfe"] and got["t1"] == ref["t1"]
    assert np.float32(got["reg_val"]) == np.float32(ref["reg_val"])
    for k in ("nf", "naccept", "nreject"):
        assert got["stats"][k] == ref["stats"][k]
    if mode != "none":
        assert got["nfe"] == ref["stats"]["nf"] + 9 and got["reg_val"] > 0


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
@pytest.mark.parametrize("mode,w_reg", [("none", 0.0), ("unbiased", 1.0), ("biased", 2.5)])
def test_layer_pullback_with_an_adams_solver_equals_the_oracle(oracle, gpu_pkg, solver, mode, w_reg):
    """The recorded Adams forward (Hermite interpolant in the record's polynomial form) feeds the handle's continuous adjoint;
    the oracle does the same (its dense recorder, its Tsit5 reversed solve): same bits."""
    import torch
    fld, h, p, x = _mk(oracle, gpu_pkg, 784, 100, 32, "tanh", True, scale=1.5)
    h.set_solver(solver)
    g = np.random.default_rng(4).standard_normal(x.shape).astype(np.float32)
    ref = oracle.node_backward(fld, x, 0.0, 1.0, 1e-4, 1e-4, g, mode=mode, t1_or_rand=0.43, w_reg=w_reg, solver=solver)
    got = h.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-4, 1e-4, torch.from_numpy(g).cuda(), mode=mode,
                          t1_or_rand=0.43, w_reg=w_reg, maxiters=10000)
    assert ref["retcode"] == 0
    for k in ("nf", "naccept", "nreject"):
        assert got["stats_fwd"][k] == ref["stats_fwd"][k], (k, got["stats_fwd"], ref["stats_fwd"])
    for k in ("naccept", "nreject", "nf", "iters", "dt_init", "t_final", "eest_last"):
        assert got["stats_bwd"][k] == ref["stats_bwd"][k], (k, got["stats_bwd"], ref["stats_bwd"])
    assert np.array_equal(got["dx"].cpu().numpy(), ref["dx"]) and np.array_equal(got["dp"].cpu().numpy(), ref["dp"])


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_adams_pullback_against_float64_autograd(oracle, gpu_pkg, solver):
    """The gradient of <g, u(1)> by the continuous adjoint over the Adams forward's Hermite record, against torch float64
    autograd through a fine RK4 of the same field (independent of both restatements)."""
    import torch
    D, H, B = 8, 16, 6
    fld, h, p, x = _mk(oracle, gpu_pkg, D, H, B, "tanh", True, scale=2.0)
    h.set_solver(solver)
    g = np.random.default_rng(4).standard_normal(x.shape).astype(np.float32)
    got = h.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, 1e-6, 1e-6, torch.from_numpy(g).cuda(), mode="none", maxiters=100000)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)

    def f(y, t):
        W1 = pt[:H * (D + 1)].reshape(D + 1, H); b1 = pt[H * (D + 1):H * (D + 1) + H]
        o2 = H * (D + 1) + H
        W2 = pt[o2:o2 + D * (H + 1)].reshape(H + 1, D); b2 = pt[o2 + D * (H + 1):]
        tt = torch.full((B, 1), t, dtype=torch.float64)
        hh = torch.tanh(torch.cat([y, tt], 1) @ W1 + b1)
        return torch.cat([hh, tt], 1) @ W2 + b2
    N = 400
    y = xt
    for i in range(N):
        t = i / N; dt = 1.0 / N
        a = f(y, t); b = f(y + 0.5 * dt * a, t + 0.5 * dt); c = f(y + 0.5 * dt * b, t + 0.5 * dt); d = f(y + dt * c, t + dt)
        y = y + dt / 6 * (a + 2 * b + 2 * c + d)
    (y * torch.tensor(g, dtype=torch.float64)).sum().backward()
    ex, ep = xt.grad.numpy(), pt.grad.numpy()
    rx = np.linalg.norm(got["dx"].cpu().numpy() - ex) / np.linalg.norm(ex)
    rp = np.linalg.norm(got["dp"].cpu().numpy() - ep) / np.linalg.norm(ep)
    print(solver, "pullback vs float64 autograd: dx %.2e dp %.2e" % (rx, rp))
    # the forward is an order-3 method at tol 1e-6 (global error ~1e-5 of the state): the gradient inherits that
    assert rx < 2e-4 and rp < 2e-4


def test_layer_object_takes_the_solver_name(gpu_pkg):
    """`NeuralODE(model; solver=VCAB3())` — the layer object hands the choice to its handle; an unknown name is the
    ArgumentError of experiments/src/construct.jl:163"""
    import torch
    P = gpu_pkg
    model = P.TDChain(P.Chain(P.Dense(5, 8, "tanh"), P.Dense(9, 4)))
    ps = torch.from_numpy(P.glorot_params(model, seed=1)).cuda()
    x = torch.rand(7, 4, device="cuda")
    outs = {}
    for s in ("Tsit5", "VCAB3()", "vcabm3"):
        node = P.NeuralODE(model, solver=s, regularize="unbiased", abstol=1e-5, reltol=1e-5)
        st = node.initialstates(np.random.default_rng(0))
        sol, st2 = node(x, ps, st)
        outs[s] = (sol.u[-1].cpu().numpy(), st2["nfe"], float(st2["reg_val"]))
        assert st2["nfe"] > 9 and st2["reg_val"] > 0
    assert np.abs(outs["VCAB3()"][0] - outs["Tsit5"][0]).max() < 1e-3 and np.abs(outs["vcabm3"][0] - outs["Tsit5"][0]).max() < 1e-3
    assert not np.array_equal(outs["VCAB3()"][0], outs["Tsit5"][0])
    with pytest.raises(ValueError):
        P.NeuralODE(model, solver="rk4")


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_training_step_with_an_adams_solver(oracle, gpu_pkg, solver):
    """run_training_step (experiments/src/utils.jl:104-123) with n.solver = VCAB3() / VCABM3(): the fused forward + head
    call falls back to its two-call form (the Adams solve has no last launch to enqueue the head behind) and the recorded
    backward runs over the Hermite record; loss and gradients against the oracle with the same t1 draw."""
    import copy
    import torch
    P, O = gpu_pkg, oracle
    D, H, B, K = 40, 24, 16, 10
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    node = P.NeuralODE(model, solver=solver, regularize="unbiased", abstol=1e-5, reltol=1e-5, save_start=False, maxiters=4000)
    rng = np.random.default_rng(8)
    ps = (P.glorot_params(model, seed=2) * np.float32(2.0)).astype(np.float32)
    pc = (rng.standard_normal(K * (D + 1)) * 0.2).astype(np.float32)
    x = rng.standard_normal((B, D)).astype(np.float32)
    lab = rng.integers(0, K, B).astype(np.int32)
    st = node.initialstates(np.random.default_rng(0))
    w_reg = 2.5
    loss, st_, stats, grads, times = P.run_training_step(node, torch.from_numpy(ps).cuda(), torch.from_numpy(pc).cuda(), st,
                                                         torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), w_reg)
    t1 = np.float32(np.float32(copy.deepcopy(st["rng"]).random(dtype=np.float32)) * 1.0 + 0.0)
    fld = O.MlpField(D, H, ps, nthreads=4)
    fo = O.node_forward(fld, x, 0.0, 1.0, 1e-5, 1e-5, mode="unbiased", t1_or_rand=t1, maxiters=4000, solver=solver)
    lo, lg, du, dpc = O.classifier_ce(fo["u_end"], pc, K, lab)
    assert st_["nfe"] == fo["nfe"]
    assert abs(float(loss) - (float(lo) + w_reg * float(fo["reg_val"]))) <= 1e-5 * abs(float(loss))
    bo = O.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, du, mode="unbiased", t1_or_rand=t1, w_reg=w_reg, maxiters=4000, solver=solver)
    gp, gx = grads["neural_ode"].cpu().numpy(), grads["x"].cpu().numpy()
    assert np.abs(gp - bo["dp"]).max() <= 2e-3 * np.abs(bo["dp"]).max()
    assert np.abs(gx - bo["dx"]).max() <= 2e-3 * np.abs(bo["dx"]).max()


def test_time_series_layer_with_an_adams_solver(gpu_pkg):
    """user saveat (the Latent-ODE caller's shape, experiments/src/construct.jl:244-249) with VCAB3: the series comes from the
    Hermite interpolant, the pullback takes a cotangent on every saved state; against float64 autograd through a fine RK4"""
    import torch
    P = gpu_pkg
    D, H, B = 6, 12, 5
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    sv = [0.25, 0.5, 0.8, 1.0]
    node = P.NeuralODE(model, solver="vcab3", regularize="none", abstol=1e-6, reltol=1e-6, saveat=sv, save_start=False, maxiters=100000)
    p = (P.glorot_params(model, seed=3) * np.float32(2.0)).astype(np.float32)
    x = np.random.default_rng(1).standard_normal((B, D)).astype(np.float32)
    st = node.initialstates(np.random.default_rng(0))
    ps = torch.from_numpy(p).cuda()
    sol, st2 = node(torch.from_numpy(x).cuda(), ps, st)
    assert [float(t) for t in sol.t] == [np.float32(t) for t in sv]
    gs = [np.random.default_rng(10 + i).standard_normal((B, D)).astype(np.float32) for i in range(len(sv))]
    dx, dp, info = node.pullback(torch.from_numpy(x).cuda(), ps, st, torch.from_numpy(np.stack(gs)).cuda())
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)

    def f(y, t):
        W1 = pt[:H * (D + 1)].reshape(D + 1, H); b1 = pt[H * (D + 1):H * (D + 1) + H]
        o2 = H * (D + 1) + H
        W2 = pt[o2:o2 + D * (H + 1)].reshape(H + 1, D); b2 = pt[o2 + D * (H + 1):]
        tt = torch.full((B, 1), t, dtype=torch.float64)
        hh = torch.tanh(torch.cat([y, tt], 1) @ W1 + b1)
        return torch.cat([hh, tt], 1) @ W2 + b2
    N = 400
    y, loss, nxt = xt, 0.0, 0
    for i in range(N):
        t = i / N; dt = 1.0 / N
        a = f(y, t); b = f(y + 0.5 * dt * a, t + 0.5 * dt); c = f(y + 0.5 * dt * b, t + 0.5 * dt); d = f(y + dt * c, t + dt)
        y = y + dt / 6 * (a + 2 * b + 2 * c + d)
        if nxt < len(sv) and abs((i + 1) / N - sv[nxt]) < 1e-9:
            err = np.abs(sol.u[nxt].cpu().numpy() - y.detach().numpy()).max()
            assert err < 5e-5, (nxt, err)
            loss = loss + (y * torch.tensor(gs[nxt], dtype=torch.float64)).sum(); nxt += 1
    assert nxt == len(sv)
    loss.backward()
    rx = np.linalg.norm(dx.cpu().numpy() - xt.grad.numpy()) / np.linalg.norm(xt.grad.numpy())
    rp = np.linalg.norm(dp.cpu().numpy() - pt.grad.numpy()) / np.linalg.norm(pt.grad.numpy())
    print("time series, vcab3: dx %.2e dp %.2e" % (rx, rp))
    assert rx < 2e-4 and rp < 2e-4


def test_solver_code_is_validated_and_the_conv_field_keeps_tsit5(gpu_pkg):
    """lrnde_set_solver rejects anything but 0 / 1 / 2 (LRNDE_BADARG, the handle stays usable); the conv field's handle has no
    Adams solve (csrc/lrnde_adams.hpp is built on the MLP handle's vector helpers): the layer object says so when it is made."""
    import ctypes as C
    import torch
    from localregneuralde_jl_amd import _lib as L
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    P = gpu_pkg
    model = P.TDChain(P.Chain(P.Dense(5, 8, "tanh"), P.Dense(9, 4)))
    h = Handle(_mlp_desc(model))
    assert L.lib.lrnde_set_solver(h._ctx, 3) == 4 and L.lib.lrnde_set_solver(h._ctx, -1) == 4
    h.set_solver("vcab3"); h.set_solver("Tsit5()")
    h.set_params(torch.from_numpy(P.glorot_params(model, seed=1)))
    assert h.rhs(torch.rand(3, 4, device="cuda"), 0.2).shape == (3, 4)
    conv = P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")), P.Chain(P.Conv((3, 3), 65, 64), P.BatchNorm(64, "gelu")),
                             P.Conv((3, 3), 65, 8)))
    P.NeuralODE(conv, solver="Tsit5")
    with pytest.raises(NotImplementedError):
        P.NeuralODE(conv, solver="vcab3")
