"""The oracle's VCAB3 / VCABM3 restatement (oracle/lrnde_oracle.c adams_solve) against the mathematics it claims to be —
UPSTREAM-RECALL: OrdinaryDiffEq's source for these methods is not in the reference, so this is the pin there is:
  * every multistep step equals the INTEGRAL OF THE INTERPOLATING POLYNOMIAL through the past derivatives, written here in
    float64 Lagrange form (no phi / phi* / beta / g recurrences: a different formula for the same method);
  * convergence to a closed-form solution as the tolerance shrinks; the evaluation count of each method."""
import numpy as np
import pytest


def _lagrange_integral(ts, fs, a, b):
    """integral over [a, b] of the polynomial through (ts[i], fs[i]) — float64, Gauss-Legendre (exact for degree <= 5)"""
    xs, ws = np.polynomial.legendre.leggauss(4)
    mid, half = 0.5 * (a + b), 0.5 * (b - a)
    tot = 0.0
    for x, w in zip(xs, ws):
        tq = mid + half * x
        val = 0.0
        for i in range(len(ts)):
            li = 1.0
            for j in range(len(ts)):
                if j != i:
                    li *= (tq - ts[j]) / (ts[i] - ts[j])
            val = val + li * fs[i]
        tot = tot + w * val
    return half * tot


def _field():
    A = np.array([[-0.5, 2.0, 0.0], [-2.0, -0.5, 0.3], [0.1, 0.0, -1.0]], np.float64)

    def f64(u, t):
        return u @ A.T + np.sin(3.0 * t) * np.array([1.0, 0.5, -0.3])
    return f64


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_multistep_steps_are_the_integral_of_the_interpolating_polynomial(oracle, solver):
    f64 = _field()
    fld = oracle.PyField(3, lambda u, t: f64(u.astype(np.float64), float(t)).astype(np.float32))
    u0 = np.array([[1.0, 2.0, -1.0], [0.5, 0.1, 3.0]], np.float32)
    r = oracle.solve(fld, u0, 0.0, 2.0, 1e-5, 1e-5, solver=solver)   # every accepted step saved
    assert r["retcode"] == 0 and r["stats"]["naccept"] > 12
    t = np.concatenate([[0.0], r["t"].astype(np.float64)])
    u = np.concatenate([u0[None].astype(np.float64), r["u"].astype(np.float64)])
    f = np.stack([f64(u[i], t[i]) for i in range(len(t))])
    worst = 0.0
    for n in range(2, len(t) - 1):   # steps 0 and 1 are the Bogacki-Shampine start
        if solver == "vcab3":   # quadratic through f_{n-2}, f_{n-1}, f_n, integrated over the step
            un1 = u[n] + _lagrange_integral(t[n - 2:n + 1], f[n - 2:n + 1], t[n], t[n + 1])
        else:   # predictor: the line through f_{n-1}, f_n; corrector: the quadratic through f_{n-1}, f_n, f(predicted)
            up = u[n] + _lagrange_integral(t[n - 1:n + 1], f[n - 1:n + 1], t[n], t[n + 1])
            fp = f64(up, t[n + 1])
            un1 = u[n] + _lagrange_integral(np.array([t[n - 1], t[n], t[n + 1]]), np.stack([f[n - 1], f[n], fp]), t[n], t[n + 1])
        worst = max(worst, np.abs(un1 - u[n + 1]).max() / max(1.0, np.abs(u[n + 1]).max()))
    print(solver, "largest deviation of a step from the polynomial-integral form:", worst)
    assert worst < 5e-6   # float32 rounding of one step (the states are O(1))


@pytest.mark.parametrize("solver", ["vcab3", "vcabm3"])
def test_adams_converges_to_the_closed_form_solution(oracle, solver):
    fld = oracle.PyField(3, lambda u, t: (-u + np.sin(np.float32(3) * t)).astype(np.float32))
    u0 = np.array([[1.0, 2.0, -1.0], [0.5, 0.1, 3.0]], np.float32)

    def exact(t):
        return (u0 + 0.3) * np.exp(-t) + (np.sin(3 * t) - 3 * np.cos(3 * t)) / 10
    errs, nfs = [], []
    for tol in (1e-3, 1e-4, 1e-5, 1e-6):
        r = oracle.solve(fld, u0, 0.0, 2.0, tol, tol, saveat=[0.7, 2.0], solver=solver)
        assert r["retcode"] == 0 and np.array_equal(r["t"], np.array([0.7, 2.0], np.float32))
        errs.append(max(np.abs(r["u"][i] - exact(tt)).max() for i, tt in enumerate(r["t"])))   # 0.7 is interpolated (Hermite)
        nfs.append(r["stats"]["nf"])
        per = 2 if solver == "vcabm3" else 1
        nstart, acc = 0, 0
        for row in r["trace"]:
            if acc < 2:
                nstart += 1
            acc += int(row["accepted"])
        assert r["stats"]["nf"] == 3 + 3 * nstart + per * (len(r["trace"]) - nstart)
    print(solver, ["%.1e" % e for e in errs], nfs)
    assert errs[0] < 5e-3 and errs[-1] < 5e-5 and all(errs[i + 1] < errs[i] for i in range(3))
    assert all(nfs[i + 1] > nfs[i] for i in range(3))


def test_unknown_solver_and_tstops_are_rejected(oracle):
    fld = oracle.PyField(1, lambda u, t: -u)
    o = oracle.make_opts(1e-3, 1e-3)
    o.alg = 7
    st = oracle.Stats()
    u0 = np.ones((1, 1), np.float32)
    us = np.empty((4, 1, 1), np.float32); ts = np.empty(4, np.float32)
    import ctypes as C
    rc = oracle.lib().lro_solve(C.byref(fld.field), u0.ctypes.data_as(C.POINTER(C.c_float)), 1, 0.0, 1.0, C.byref(o), None, 0,
                                us.ctypes.data_as(C.POINTER(C.c_float)), ts.ctypes.data_as(C.POINTER(C.c_float)), 4, C.byref(st), None, 0)
    assert rc == 4
