"""Known-answer tests that pin the CPU oracle (the reference's own tests hold no numeric
vectors — SURVEY.md §4 — so these are the pins): tableau identities, convergence order,
an independent numpy restatement, scipy, and the committed golden fixtures."""
import itertools
import os

import numpy as np
import pytest

from np_restatement import NpMlp, tsit5_step as np_step

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _butcher(O):
    a, c, bt, r = O.tableau()
    Am = np.zeros((7, 7))
    idx = 0
    for i in range(1, 7):
        for j in range(i):
            Am[i, j] = a[idx]; idx += 1
    cv = np.concatenate([[0.0], c])
    b = Am[6].copy()
    return Am, cv, b, bt, r


def test_tableau_row_sums_and_fsal(oracle):
    Am, c, b, bt, r = _butcher(oracle)
    assert np.allclose(Am.sum(axis=1), c, atol=1e-15)
    assert abs(bt.sum()) < 1e-16
    assert c[6] == 1.0 and b[6] == 0.0


def test_tableau_order_conditions_through_5(oracle):
    """All 17 rooted-tree conditions up to order 5 for b = a7.."""
    Am, c, b, bt, r = _butcher(oracle)
    Ac = Am @ c
    conds = [
        (b.sum(), 1), (b @ c, 1 / 2), (b @ c ** 2, 1 / 3), (b @ Ac, 1 / 6),
        (b @ c ** 3, 1 / 4), (b @ (c * Ac), 1 / 8), (b @ (Am @ c ** 2), 1 / 12), (b @ (Am @ Ac), 1 / 24),
        (b @ c ** 4, 1 / 5), (b @ (c ** 2 * Ac), 1 / 10), (b @ (Ac * Ac), 1 / 20),
        (b @ (c * (Am @ c ** 2)), 1 / 15), (b @ (Am @ c ** 3), 1 / 20), (b @ (c * (Am @ Ac)), 1 / 30),
        (b @ (Am @ (c * Ac)), 1 / 40), (b @ (Am @ (Am @ c ** 2)), 1 / 60), (b @ (Am @ (Am @ Ac)), 1 / 120)]
    for i, (lhs, rhs) in enumerate(conds):
        assert abs(lhs - rhs) < 5e-15, (i, lhs, rhs)
    # the embedded weights b - btilde satisfy the conditions through order 4
    bh = b - bt
    for lhs, rhs in [(bh.sum(), 1), (bh @ c, 1 / 2), (bh @ c ** 2, 1 / 3), (bh @ Ac, 1 / 6), (bh @ c ** 3, 1 / 4),
                     (bh @ (c * Ac), 1 / 8), (bh @ (Am @ c ** 2), 1 / 12), (bh @ (Am @ Ac), 1 / 24)]:
        assert abs(lhs - rhs) < 5e-15


def test_dense_output_identities(oracle):
    Am, c, b, bt, r = _butcher(oracle)
    def bth(th):
        out = np.empty(7)
        out[0] = th * (r[0, 0] + th * (r[0, 1] + th * (r[0, 2] + th * r[0, 3])))
        for i in range(1, 7):
            out[i] = th * th * (r[i, 1] + th * (r[i, 2] + th * r[i, 3]))
        return out
    assert np.allclose(bth(1.0), b, atol=2e-14)
    Ac = Am @ c
    for th in (0.25, 0.5, 0.8):
        w = bth(th)
        assert abs(w.sum() - th) < 1e-13
        assert abs(w @ c - th ** 2 / 2) < 1e-13
        assert abs(w @ c ** 2 - th ** 3 / 3) < 1e-13
        assert abs(w @ Ac - th ** 3 / 6) < 1e-13
        assert abs(w @ c ** 3 - th ** 4 / 4) < 1e-13


def test_local_error_order_and_eest_scaling(oracle):
    """u' = -u: local error at least O(dt^6) (Tsit5's dt^6 coefficient is tiny on linear problems,
    so the observed ratio sits between 2^6 and 2^7), embedded estimate ~ dt^5 (5(4) pair)."""
    f = oracle.PyField(4, lambda u, t: -u)
    u0 = np.array([[1.0, 2.0, -0.5, 0.25]], dtype=np.float32)
    errs, ees = [], []
    for dt in (0.8, 0.4):
        r = oracle.tsit5_step(f, u0, -u0, 0.0, dt, 1e-3, 1e-3)
        errs.append(np.abs(r["u"].astype(np.float64) - u0 * np.exp(-dt)).max())
        ees.append(float(r["eest"]))
    assert 60 < errs[0] / errs[1] < 135, errs     # between 2^6 and 2^7
    assert 25 < ees[0] / ees[1] < 50, ees         # ~2^5 (scaled by the |u| weights)


def test_harmonic_oscillator_solve_matches_scipy(oracle):
    from scipy.integrate import solve_ivp
    f = oracle.PyField(2, lambda u, t: np.stack([u[:, 1], -u[:, 0]], axis=1))
    u0 = np.array([[1.0, 0.0], [0.3, -0.7]], dtype=np.float32)
    s = oracle.solve(f, u0, 0.0, 3.0, 1e-6, 1e-6, saveat=[1.0, 3.0], maxiters=1000)
    assert s["retcode"] == 0 and s["stats"]["nf"] == 3 + 6 * (s["stats"]["naccept"] + s["stats"]["nreject"])
    for row in range(2):
        ref = solve_ivp(lambda t, y: [y[1], -y[0]], (0, 3), u0[row].astype(float), rtol=1e-10, atol=1e-12,
                        t_eval=[1.0, 3.0])
        assert np.allclose(s["u"][:, row, :], ref.y.T, atol=2e-5)


def test_saveat_interpolation_accuracy(oracle):
    """saveat points are filled by the 4th-order Tsit5 interpolant, not by stepping to them."""
    f = oracle.PyField(1, lambda u, t: -u)
    u0 = np.ones((1, 1), np.float32)
    sv = np.linspace(0.05, 2.0, 40).astype(np.float32)
    s = oracle.solve(f, u0, 0.0, 2.0, 1e-5, 1e-5, saveat=sv, maxiters=1000)
    assert s["stats"]["naccept"] < 20 and len(s["t"]) == 40
    assert np.allclose(s["u"][:, 0, 0], np.exp(-sv.astype(np.float64)), atol=5e-5)


@pytest.mark.parametrize("D,H,B,act,td,dt", [(784, 100, 8, "tanh", True, 0.1), (2, 4, 1, "gelu", True, 0.6),
                                             (20, 40, 5, "gelu", False, 0.3)])
def test_numpy_restatement_agrees(oracle, D, H, B, act, td, dt):
    # weights x6: the truncation error must dominate fp32 rounding noise in utilde, otherwise
    # EEst is noise and two correct implementations legitimately disagree
    p = oracle.glorot_mlp_params(D, H, time_dep=td, seed=3) * np.float32(6.0)
    p[:] += np.random.default_rng(4).standard_normal(p.size).astype(np.float32) * 0.01
    x = np.random.default_rng(5).random((B, D), dtype=np.float32)
    fo = oracle.MlpField(D, H, p, time_dep=td, act=act)
    fn = NpMlp(D, H, p, time_dep=td, act=act)
    def close(a, b, rtol):  # rtol relative to the array scale (dot products cancel)
        return np.abs(a.astype(np.float64) - b).max() <= rtol * np.abs(b).max()
    assert close(fo.rhs(x, 0.3), fn(x, np.float32(0.3)), 1e-5)
    k1 = fo.rhs(x, 0.1)
    a = oracle.tsit5_step(fo, x, k1, 0.1, dt, 1e-3, 1e-3)
    b = np_step(fn, x, k1, 0.1, dt, 1e-3, 1e-3)
    assert close(a["u"], b["u"], 1e-5)
    assert close(a["k7"], b["k7"], 1e-5)
    for k in ("eest", "reg_error", "reg_stiff"):
        assert np.isclose(a[k], b[k], rtol=2e-3), (k, a[k], b[k])


def test_activations_accuracy(oracle):
    x = np.concatenate([np.linspace(-12, 12, 4001), np.linspace(-0.7, 0.7, 2001)]).astype(np.float32)
    t = oracle.vec_fn("lro_tanhf", x)
    ref = np.tanh(x.astype(np.float64))
    assert np.max(np.abs(t - ref) / np.maximum(np.abs(ref), 1e-30)) < 2.5e-7
    g = oracle.vec_fn("lro_geluf", x)
    xd = x.astype(np.float64)
    refg = 0.5 * xd * (1 + np.tanh(np.sqrt(2 / np.pi) * (xd + 0.044715 * xd ** 3)))
    assert np.max(np.abs(g - refg)) < 1e-6
    e = oracle.vec_fn("lro_expf", np.linspace(-87, 87, 3001).astype(np.float32))
    assert np.max(np.abs(e / np.exp(np.linspace(-87, 87, 3001).astype(np.float32).astype(np.float64)) - 1)) < 2e-7


def test_fastpow_is_the_diffeqbase_approximation(oracle):
    L = oracle.lib()
    for x, y in itertools.product((1e-4, 0.03, 0.5, 1.0, 3.7, 250.0), (0.14, 0.08)):
        got = L.lro_fastpow(x, y)
        assert abs(got / x ** y - 1) < 2e-3          # documented relative error ~1e-4..1e-3
    assert L.lro_fastpow(0.0, 0.14) == 0.0
    assert L.lro_fastpow2(0.0) == pytest.approx(1.0, rel=1e-4)


def test_step_counts_formula_and_retcodes(oracle):
    p = oracle.glorot_mlp_params(16, 32, seed=0) * np.float32(20)
    x = np.random.default_rng(0).random((4, 16), dtype=np.float32)
    f = oracle.MlpField(16, 32, p)
    s = oracle.solve(f, x, 0, 1, 1e-4, 1e-4, saveat=[1.0], maxiters=10000)
    st = s["stats"]
    assert st["nf"] == 3 + 6 * (st["naccept"] + st["nreject"]) and st["nreject"] > 0
    assert np.all(s["trace"]["eest"][s["trace"]["accepted"] == 1] <= 1.0)
    assert np.all(s["trace"]["eest"][s["trace"]["accepted"] == 0] > 1.0)
    s2 = oracle.solve(f, x, 0, 1, 1e-4, 1e-4, saveat=[1.0], maxiters=3)
    assert s2["retcode"] == 1


def test_node_forward_behaviour_pins(oracle):
    """The assertions test/runtests.jl makes on the forward pass (:21-22, :118): reg_val is
    zero iff regularize == :none; nfe = sol.nf + 9 in the regularised modes."""
    D, H, B = 2, 4, 1
    p = oracle.glorot_mlp_params(D, H, seed=0)
    x = np.random.default_rng(0).standard_normal((B, D)).astype(np.float32)
    f = oracle.MlpField(D, H, p, act="gelu")
    none = oracle.node_forward(f, x, 0, 1, 1e-6, 1e-3, mode="none")
    assert none["reg_val"] == 0 and none["nfe"] == none["stats"]["nf"] and none["u_end"].dtype == np.float32
    for mode in ("unbiased", "biased"):
        for rt in ("error_estimate", "stiffness_estimate"):
            r = oracle.node_forward(f, x, 0, 1, 1e-6, 1e-3, mode=mode, reg_type=rt, t1_or_rand=0.41)
            assert r["reg_val"] != 0 and np.isfinite(r["reg_val"])
            assert r["nfe"] == r["stats"]["nf"] + 9
            assert np.array_equal(r["u_end"], none["u_end"]) or mode == "biased"


def test_golden_fixtures(oracle):
    """Regression pins generated by tests/golden/make_golden.py from this oracle."""
    g = np.load(os.path.join(GOLD, "mnist_mlp_b16.npz"))
    f = oracle.MlpField(784, 100, g["params"])
    r = oracle.tsit5_step(f, g["x"], g["k1"], float(g["t"]), float(g["dt"]), float(g["tol"]), float(g["tol"]))
    assert np.array_equal(r["u"], g["step_u"]) and np.array_equal(r["k7"], g["step_k7"])
    assert r["eest"] == g["step_eest"] and r["reg_error"] == g["step_reg_error"] and r["reg_stiff"] == g["step_reg_stiff"]
    s = oracle.solve(f, g["x"], 0.0, 1.0, float(g["tol"]), float(g["tol"]), saveat=[0.5, 1.0], maxiters=10000)
    assert np.array_equal(s["trace"]["dt"], g["solve_dt_trace"])
    assert np.array_equal(s["u"], g["solve_u"])
    assert s["stats"]["nf"] == int(g["solve_nf"])
    n = oracle.node_forward(f, g["x"], 0.0, 1.0, float(g["tol"]), float(g["tol"]), mode="unbiased", t1_or_rand=0.37,
                            maxiters=10000)
    assert n["reg_val"] == g["node_reg_val"] and n["nfe"] == int(g["node_nfe"])


def test_golden_sde_fixture(oracle):
    """tests/golden/mnist_sde_b16.npz: one Euler-Heun, one Milstein and one four-stage SRI step at the MNIST-SDE shapes"""
    g = np.load(os.path.join(GOLD, "mnist_sde_b16.npz"))
    D = 32
    p2 = np.concatenate([np.eye(D, dtype=np.float32).ravel(), np.zeros(D, np.float32), g["p_diffusion"]])
    drift = oracle.MlpField(D, 64, g["p_drift"], time_dep=False, act="tanh")
    diff = oracle.MlpField(D, D, p2, time_dep=False, act="identity")
    t, dt = float(g["t"]), g["dt"]
    eh = oracle.euler_heun_step(drift, diff, g["u"], g["dW"], t, dt, 0.14, 0.14, 1.0 / 6.0)
    assert np.array_equal(eh["u"], g["eh_u"]) and eh["eest"] == g["eh_eest"] and eh["reg_val"] == g["eh_reg"]
    rk = oracle.rkmil_step(drift, diff, g["u"], g["dW"], t, dt, 0.14, 0.14)
    assert np.array_equal(rk["u"], g["rk_u"]) and rk["eest"] == g["rk_eest"] and rk["reg_val"] == g["rk_reg"]
    sr = oracle.sri_step(drift, diff, dict(zip(oracle.SRI_FIELDS, g["tableau"].tolist())), g["u"], g["dW"], g["dZ"], t, dt, 0.14, 0.14,
                         1.0 / 6.0)
    assert np.array_equal(sr["u"], g["sri_u"]) and sr["eest"] == g["sri_eest"] and sr["reg_val"] == g["sri_reg"]


def test_rkmil_step_is_milstein_for_linear_noise(oracle):
    """dX = a X dt + b X dW (diagonal, Ito).  With K = X(1 + a dt), the derivative-free quotient of
    src/perform_step.jl:136-139 is Dg = (g(K + sqrt(dt) g(X)) - g(X))/sqrt(dt) = b^2 X + a b X sqrt(dt), so the step is
    K + b X dW + Dg (dW^2 - dt)/2 — Milstein's X(1 + a dt + b dW + b^2/2 (dW^2 - dt)) up to the O(dt^1.5) quotient
    term; EEst is the 4-argument residual norm (:218-220)."""
    O = oracle
    a, b = -0.7, 0.4
    drift = O.PyField(3, lambda u, t: np.float32(a) * u)
    diff = O.PyField(3, lambda u, t: np.float32(b) * u)
    rng = np.random.default_rng(0)
    u = rng.uniform(0.5, 2.0, (5, 3)).astype(np.float32)
    dt = np.float32(0.01)
    dW = (rng.standard_normal((5, 3)) * np.sqrt(dt)).astype(np.float32)
    r = O.rkmil_step(drift, diff, u, dW, 0.0, dt, 1e-2, 1e-2)
    u64, w64, h = u.astype(np.float64), dW.astype(np.float64), float(dt)
    expect = u64 * (1 + a * h) + b * u64 * w64 + (b * b * u64 + a * b * u64 * np.sqrt(h)) * (w64 ** 2 - h) / 2
    np.testing.assert_allclose(r["u"], expect, rtol=5e-6)
    milstein = u64 * (1 + a * h + b * w64 + 0.5 * b * b * (w64 ** 2 - h))
    assert np.abs(r["u"] - milstein).max() < 5 * abs(a * b) * h ** 1.5 * np.abs(u64).max()
    res = (r["u"].astype(np.float64) - u) / (1e-2 + np.maximum(np.abs(u), np.abs(r["u"])) * 1e-2)
    assert abs(float(r["eest"]) - np.sqrt(np.mean(res ** 2))) <= 1e-5 * float(r["eest"])
    assert r["reg_val"] == np.float32(r["eest"] * dt)


def _sri_zero_tableau():
    return {k: 0.0 for k in "a021 a031 a032 a041 a042 a043 a121 a131 a132 a141 a142 a143 b021 b031 b032 b041 b042 b043 b121 b131 b132 "
            "b141 b142 b143 c02 c03 c04 c11 c12 c13 c14 alpha1 alpha2 alpha3 alpha4 beta11 beta12 beta13 beta14 beta21 beta22 "
            "beta23 beta24 beta31 beta32 beta33 beta34 beta41 beta42 beta43 beta44".split()}


def test_sri_step_structure(oracle):
    """src/perform_step.jl:49-106 with caller-supplied coefficients.  (1) alpha1 = beta11 = 1, everything else 0 is
    Euler-Maruyama: u = uprev + dt f(uprev) + dW g(uprev), E2 = 0, E1 = dt (k1+k2+k3+k4).  (2) A random tableau against
    a float64 numpy transcription of the reference's lines on a linear field (every H, E and chi term exercised)."""
    O = oracle
    a, b = -0.7, 0.4
    drift = O.PyField(3, lambda u, t: np.float32(a) * u + np.float32(0.1 * t))
    diff = O.PyField(3, lambda u, t: np.float32(b) * u)
    rng = np.random.default_rng(1)
    u = rng.uniform(0.5, 2.0, (5, 3)).astype(np.float32)
    t, dt = np.float32(0.2), np.float32(0.01)
    dW = (rng.standard_normal((5, 3)) * np.sqrt(dt)).astype(np.float32)
    dZ = (rng.standard_normal((5, 3)) * np.sqrt(dt)).astype(np.float32)
    T = _sri_zero_tableau(); T["alpha1"] = 1.0; T["beta11"] = 1.0
    r = O.sri_step(drift, diff, T, u, dW, dZ, t, dt, 1e-2, 1e-2, 1.0 / 6.0)
    em = u + dt * (np.float32(a) * u + np.float32(0.1 * t)) + dW * (np.float32(b) * u)
    np.testing.assert_allclose(r["u"], em, rtol=2e-7)
    f0 = a * u.astype(np.float64) + 0.1 * float(t)
    E1 = float(dt) * 4 * f0  # all stages sit at uprev and time t (every a, b, c is zero)
    res = (E1 / 6.0) / (1e-2 + np.maximum(np.abs(u), np.abs(r["u"])) * 1e-2)
    assert abs(float(r["eest"]) - np.sqrt(np.mean(res ** 2))) <= 2e-6 * float(r["eest"])
    assert r["reg_val"] == np.float32(r["eest"] * dt)
    # random tableau vs a float64 transcription
    T = {k: float(v) for k, v in zip(T.keys(), rng.uniform(-0.8, 0.8, len(T)))}
    r = O.sri_step(drift, diff, T, u, dW, dZ, t, dt, 1e-2, 1e-2, 1.0 / 6.0)
    U, W, Z, h, tt = u.astype(np.float64), dW.astype(np.float64), dZ.astype(np.float64), float(dt), float(t)
    f = lambda x, s: a * x + 0.1 * s
    g = lambda x, s: b * x
    sq = np.sqrt(h)
    chi1 = (W ** 2 - h) / (2 * sq); chi2 = (W + Z / np.sqrt(3.0)) / 2; chi3 = (W ** 3 - 3 * W * h) / (6 * h)
    k1 = f(U, tt); g1 = g(U, tt + T["c11"] * h)
    H01 = U + h * T["a021"] * k1 + T["b021"] * chi2 * g1; H11 = U + h * T["a121"] * k1 + sq * T["b121"] * g1
    k2 = f(H01, tt + T["c02"] * h); g2 = g(H11, tt + T["c12"] * h)
    H02 = U + h * (T["a031"] * k1 + T["a032"] * k2) + chi2 * (T["b031"] * g1 + T["b032"] * g2)
    H12 = U + h * (T["a131"] * k1 + T["a132"] * k2) + sq * (T["b131"] * g1 + T["b132"] * g2)
    k3 = f(H02, tt + T["c03"] * h); g3 = g(H12, tt + T["c13"] * h)
    H03 = U + h * (T["a041"] * k1 + T["a042"] * k2 + T["a043"] * k3) + chi2 * (T["b041"] * g1 + T["b042"] * g2 + T["b043"] * g3)
    H13 = U + h * (T["a141"] * k1 + T["a142"] * k2 + T["a143"] * k3) + sq * (T["b141"] * g1 + T["b142"] * g2 + T["b143"] * g3)
    k4 = f(H03, tt + T["c04"] * h); g4 = g(H13, tt + T["c14"] * h)
    E2 = chi2 * (T["beta31"] * g1 + T["beta32"] * g2 + T["beta33"] * g3 + T["beta34"] * g4) + \
        chi3 * (T["beta41"] * g1 + T["beta42"] * g2 + T["beta43"] * g3 + T["beta44"] * g4)
    un = U + h * (T["alpha1"] * k1 + T["alpha2"] * k2 + T["alpha3"] * k3 + T["alpha4"] * k4) + E2 + \
        W * (T["beta11"] * g1 + T["beta12"] * g2 + T["beta13"] * g3 + T["beta14"] * g4) + \
        chi1 * (T["beta21"] * g1 + T["beta22"] * g2 + T["beta23"] * g3 + T["beta24"] * g4)
    np.testing.assert_allclose(r["u"], un, rtol=3e-6)
    E1 = h * (k1 + k2 + k3 + k4)
    res = (E1 / 6.0 + E2) / (1e-2 + np.maximum(np.abs(U), np.abs(un)) * 1e-2)
    assert abs(float(r["eest"]) - np.sqrt(np.mean(res ** 2))) <= 1e-4 * float(r["eest"])
