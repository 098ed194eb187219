"""GPU parity against an arithmetic that is NOT the co-designed oracle: tests/np_restatement.py — float64 BLAS field
rounded once to float32, float32 numpy stage arithmetic, its own initdt / PI controller / fastpow written from
SURVEY.md §3.5 (no code, no summation order shared with oracle/lrnde_oracle.c or the kernels).

Tolerances (north star: rtol 1e-5 fp32 on results, exact accepted-step counts):
  * f-eval, one Tsit5 step (u, k7), sol.u[end]: max |gpu - restatement| <= 1e-5 * max |restatement|  (scale-relative).
  * accepted / rejected step counts: EQUAL wherever the embedded error estimate is a property of the ODE — weights x3 and
    x6 of the glorot scale, where truncation error dominates.  At the glorot scale itself the estimate is fp32 rounding
    noise of the field (first step at tol 1e-4: EEst 3.3e-5 with the MFMA summation order, 4.8e-5 with OpenBLAS sgemm,
    6.1e-6 with a float64 field), so the counts there belong to a summation order, not to the algorithm: they are
    printed, and bounded, not asserted equal.  Only a run of the Julia reference can pin them (tests/test_true_reference.py).
"""
import numpy as np
import pytest

import np_restatement as R

pytestmark = pytest.mark.gpu


def _mk(pkg, D, H, B, scale, seed=0, act="tanh"):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    model = pkg.TDChain(pkg.Chain(pkg.Dense(D + 1, H, act), pkg.Dense(H + 1, D)))
    p = pkg.glorot_params(model, seed=seed) * np.float32(scale)
    p = (p + np.random.default_rng(seed + 1).standard_normal(p.size).astype(np.float32) * np.float32(0.01)).astype(np.float32)
    x = np.random.default_rng(seed + 2).random((B, D), dtype=np.float32)
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(p))
    return h, p, x, model


def _err(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / np.abs(b).max())


@pytest.mark.parametrize("D,H,B,act", [(784, 100, 512, "tanh"), (784, 100, 37, "tanh"), (32, 64, 33, "gelu")])
def test_feval_and_step_vs_float64_restatement(gpu_pkg, D, H, B, act):
    import torch
    h, p, x, _ = _mk(gpu_pkg, D, H, B, 1.0, act=act)
    f = R.NpMlp64(D, H, p, act=act)
    xd = torch.from_numpy(x).cuda()
    for t in (0.0, 0.37):
        e = _err(h.rhs(xd, t).cpu().numpy(), f(x, t))
        assert e <= 1e-5, ("f-eval", t, e)
    k1 = f(x, 0.1)
    ref = R.tsit5_step(f, x, k1, 0.1, 0.05, 1e-4, 1e-4)
    got = h.perform_step(xd, torch.from_numpy(k1).cuda(), 0.1, 0.05, 1e-4, 1e-4)
    assert _err(got["u"].cpu().numpy(), ref["u"]) <= 1e-5
    assert _err(got["k7"].cpu().numpy(), ref["k7"]) <= 1e-5
    dt_ref, f0 = R.init_dt(f, x, 0.0, 1.0, 1e-4, 1e-4)
    dt, k1g = h.init_dt(xd, 0.0, 1.0, 1e-4, 1e-4)
    assert abs(float(dt) - float(dt_ref)) <= 1e-4 * float(dt_ref)  # log10/pow of rms values that agree to ~1e-6
    assert _err(k1g.cpu().numpy(), f0) <= 1e-5


@pytest.mark.parametrize("scale,tol", [(3.0, 1e-3), (3.0, 1e-4), (3.0, 1e-5), (6.0, 1e-3), (6.0, 1e-4), (6.0, 1e-5)])
def test_solve_step_counts_equal_where_truncation_dominates(gpu_pkg, scale, tol):
    """exact accepted/rejected counts and 1e-5 results against BOTH restatement fields (float64 BLAS and float32 BLAS)"""
    import torch
    D, H, B = 784, 100, 64
    h, p, x, _ = _mk(gpu_pkg, D, H, B, scale)
    got = h.solve(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, saveat=[1.0], maxiters=10000, trace=True)
    for name, f in (("float64 field", R.NpMlp64(D, H, p)), ("float32 BLAS field", R.NpMlp(D, H, p))):
        ref = R.solve(f, x, 0.0, 1.0, tol, tol)
        assert (got["stats"]["naccept"], got["stats"]["nreject"], got["stats"]["nf"]) == (ref["naccept"], ref["nreject"], ref["nf"]), name
        assert np.allclose(got["trace"]["dt"], ref["dts"], rtol=0.1), name   # same controller; dt ~ EEst^-0.14, EEst equal to a few %
        # two step sequences that differ by a few % in dt differ by the solver's own global error: the 1e-5 bar applies
        # where the solve tolerance is below it
        assert _err(got["u"][-1].cpu().numpy(), ref["u"]) <= max(1e-5, 0.5 * tol), name


@pytest.mark.parametrize("B,tol", [(512, 1.4e-8), (512, 1e-4), (64, 1e-6)])
def test_metric_configuration_solution_vs_float64_restatement(gpu_pkg, B, tol):
    """MNIST-ODE B=512 at the experiment's tolerance: sol.u[end] of the HIP path within 1e-5 of the float64-field
    restatement's (both are converged far below that); step counts differ with the summation order (module docstring)"""
    import torch
    D, H = 784, 100
    h, p, x, _ = _mk(gpu_pkg, D, H, B, 1.0)
    got = h.node_forward(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, mode="none", maxiters=10000)
    r64 = R.solve(R.NpMlp64(D, H, p), x, 0.0, 1.0, tol, tol)
    r32 = R.solve(R.NpMlp(D, H, p), x, 0.0, 1.0, tol, tol)
    e = _err(got["u_end"].cpu().numpy(), r64["u"])
    print(f"B={B} tol={tol:g}: accepted steps gpu {got['stats']['naccept']}, float32-BLAS restatement {r32['naccept']}, "
          f"float64-field restatement {r64['naccept']}; sol.u[end] err {e:.2e} of scale")
    assert e <= 1e-5
    assert _err(got["u_end"].cpu().numpy(), r32["u"]) <= 1e-5
    # noise-dominated estimates inflate the count over the float64 field's, never deflate it, and stay within 2x of it
    assert r64["naccept"] <= got["stats"]["naccept"] <= 2 * r64["naccept"] + 2
    assert got["stats"]["nreject"] <= 2


def test_corrected_solution_with_user_saveat(gpu_pkg):
    """`(n::NeuralODE{:unbiased})(x, ps, st)` with a user `saveat`: t1 is appended to saveat for the solve and dropped
    again by _CorrectedDESolution (src/layers/neural_ode.jl:107-111, src/utils.jl:25-33): the returned solution holds
    exactly the user's times, its states equal the plain saveat solve's, reg_val / nfe are those of the local step at t1."""
    import copy
    import torch
    P = gpu_pkg
    D, H, B = 784, 100, 24
    h, p, x, model = _mk(P, D, H, B, 2.0)
    xd, ps = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    user = [0.25, 0.5, 1.0]
    node = P.NeuralODE(model, regularize="unbiased", abstol=1e-5, reltol=1e-5, saveat=user, save_start=False, maxiters=1000)
    st = node.initialstates(np.random.default_rng(5))
    sol, st2 = node(xd, ps, st)
    assert [float(t) for t in sol.t] == [float(np.float32(t)) for t in user]          # t1 is gone
    assert len(sol.u) == 3
    rng = copy.deepcopy(st["rng"])
    t1 = np.float32(rng.random(dtype=np.float32))
    assert all(abs(float(t1) - u) > 1e-6 for u in user)
    plain = h.solve(xd, 0.0, 1.0, 1e-5, 1e-5, saveat=user, save_everystep=False, maxiters=1000)
    for a, b in zip(sol.u, plain["u"]):  # saveat points are interpolated, not tstops: the extra one changes nothing
        assert torch.equal(a, b)
    assert sol.destats.nf == plain["stats"]["nf"] and st2["nfe"] == plain["stats"]["nf"] + 9
    # the local step is the one at (sol(t1), t1): same reg_val as the C-side layer forward with that t1
    ref = h.node_forward(xd, 0.0, 1.0, 1e-5, 1e-5, mode="unbiased", t1_or_rand=float(t1), maxiters=1000)
    assert st2["reg_val"] == ref["reg_val"] and st2["reg_val"] > 0
    ts = P.diffeqsol_to_timeseries(sol)
    assert ts.shape == (3, B, D) and torch.equal(P.diffeqsol_to_array(sol), sol.u[-1])
    # a user saveat that CONTAINS t1 exactly: the reference's filter `t1 .!= sol.t` drops the user's own point too
    node2 = P.NeuralODE(model, regularize="unbiased", abstol=1e-5, reltol=1e-5, saveat=[float(t1), 1.0], save_start=False)
    sol2, _ = node2(xd, ps, st)
    assert [float(t) for t in sol2.t] == [1.0]
    # test mode: plain solve on the user's saveat, reg_val 0
    sol3, st3 = node(xd, ps, dict(st, training=False))
    assert st3["reg_val"] == 0 and len(sol3.u) == 3 and st3["nfe"] == plain["stats"]["nf"]


def test_bind_repacks_parameters_updated_in_place(gpu_pkg):
    """a numpy-side in-place update of the parameter storage bumps neither data_ptr nor torch's version
    counter; the layer must still see the new values"""
    import torch
    P = gpu_pkg
    D, H, B = 32, 64, 8
    h, p, x, model = _mk(P, D, H, B, 1.0)
    node = P.NeuralODE(model, regularize="none", abstol=1e-5, reltol=1e-5)
    st = node.initialstates(np.random.default_rng(0))
    pa = p.copy()
    ps = torch.from_numpy(pa)  # CPU tensor sharing memory with the numpy array
    xd = torch.from_numpy(x).cuda()
    y0 = node(xd, ps, st)[0].u[-1].clone()
    v = ps._version
    pa *= np.float32(1.5)      # in place through numpy
    assert ps._version == v
    y1 = node(xd, ps, st)[0].u[-1]
    assert not torch.equal(y0, y1)


def test_call_and_pullback_share_the_t1_draw(gpu_pkg):
    """`node(x, ps, st)` and `node.pullback(...)` (and run_training_step) pick the same t1 from the same
    st['rng'] in :biased mode (one uniform draw, index floor(r*m)), so reg_val belongs to the differentiated step"""
    import torch
    P = gpu_pkg
    D, H, B = 32, 64, 8
    h, p, x, model = _mk(P, D, H, B, 2.0)
    xd, ps = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    for seed in range(4):
        node = P.NeuralODE(model, regularize="biased", abstol=1e-4, reltol=1e-4)
        st = dict(node.initialstates(np.random.default_rng(0)), rng=np.random.default_rng(seed))
        sol, st2 = node(xd, ps, st)
        g = torch.ones_like(xd)
        dx, dp, info = node.pullback(xd, ps, st, g, w_reg=1.0)
        assert info["reg_val"] == st2["reg_val"], (seed, info["reg_val"], st2["reg_val"])
        assert float(info["t1"]) in [float(t) for t in sol.t[:-1]]
