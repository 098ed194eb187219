"""The numpy restatement (oracle/np_restatement.py) and the C oracle are two independent restatements of the same path:
they must agree within fp32 tolerance on results and exactly on the pieces that are integer / bit arithmetic.  Also
records two findings that bound what any fp32 implementation — the reference's included — can be compared at."""
import numpy as np
import pytest
import torch

import np_restatement as R


def test_fastpow_restatements_agree_bitwise(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([10.0 ** rng.uniform(-6, 2, 400), [1.0, 0.5, 2.0, 1e-4]]).astype(np.float32)
    for x in xs:
        for y in (np.float32(7.0 / 50.0), np.float32(2.0 / 25.0)):
            assert R.fastpow(x, y) == np.float32(L.lro_fastpow(float(x), float(y))), (x, y)
            assert abs(float(R.fastpow(x, y)) / float(x) ** float(y) - 1) < 5e-4   # the approximation's own accuracy


@pytest.mark.parametrize("fast", [False, True])
def test_numpy_forward_pass_matches_the_oracle(oracle, fast):
    D, H, B = 784, 100, 24
    p = oracle.glorot_mlp_params(D, H, seed=0) * np.float32(3.0)
    x = np.random.default_rng(0).random((B, D), dtype=np.float32)
    fld = oracle.MlpField(D, H, p, nthreads=4)
    ro = oracle.node_forward(fld, x, 0.0, 1.0, 1e-4, 1e-4, mode="unbiased", reg_type="error_estimate", t1_or_rand=0.37,
                             maxiters=10000)
    rn = R.node_forward(R.NpMlp(D, H, p), x, 0.0, 1.0, 1e-4, 1e-4, 0.37, fast=fast)
    assert rn["nfe"] == ro["nfe"] and rn["naccept"] == ro["stats"]["naccept"]   # truncation-dominated: counts are robust
    assert np.abs(rn["u_end"] - ro["u_end"]).max() <= 1e-5 * np.abs(ro["u_end"]).max()
    # reg_val = EEst*dt of the local step carries the field's fp32 rounding noise (next test): percent-level agreement only
    assert abs(float(rn["reg_val"]) - float(ro["reg_val"])) <= 1e-1 * float(ro["reg_val"])


def test_error_estimate_is_rounding_noise_at_the_glorot_scale(oracle):
    """Finding: for the MNIST field at its initialisation scale the embedded error estimate of a step is
    dominated by the fp32 rounding noise of the field, already at tol 1e-4 — a float64 evaluation of the same field gives
    an estimate several times smaller than either fp32 summation order.  Accepted-step counts at such tolerances are a
    property of the summation order (MFMA chain / OpenBLAS kernel), not of the algorithm."""
    D, H, B = 784, 100, 16
    p = oracle.glorot_mlp_params(D, H, seed=0)
    x = np.random.default_rng(0).random((B, D), dtype=np.float32)
    fld = oracle.MlpField(D, H, p, nthreads=4)
    k1 = fld.rhs(x, 0.0)
    dt = oracle.init_dt(fld, x, 0.0, 1.0, 1e-4, 1e-4)[0]
    e_chain = float(oracle.tsit5_step(fld, x, k1, 0.0, dt, 1e-4, 1e-4)["eest"])
    e_blas = float(R.tsit5_step(R.NpMlp(D, H, p), x, k1, 0.0, dt, 1e-4, 1e-4)["eest"])
    e_f64 = float(R.tsit5_step(R.NpMlp64(D, H, p), x, k1, 0.0, dt, 1e-4, 1e-4)["eest"])
    assert e_f64 < 0.5 * min(e_chain, e_blas), (e_chain, e_blas, e_f64)
    # ... while sol.u[end] is insensitive to it
    a = R.solve(R.NpMlp64(D, H, p), x, 0.0, 1.0, 1e-4, 1e-4)
    b = oracle.solve(fld, x, 0.0, 1.0, 1e-4, 1e-4, saveat=[1.0], maxiters=1000)
    assert np.abs(a["u"] - b["u"][-1]).max() <= 1e-5 * np.abs(a["u"]).max()


def test_reg_gradient_conditioning_in_fp32():
    """Finding behind the backward tolerances (tests/test_gpu_backward.py): d reg_val / d ps of the :error_estimate
    regulariser differentiates utilde = dt * sum btilde_j k_j, a sum that cancels to ~1e-4 of its terms (sum btilde = 0),
    so ANY float32 evaluation — here plain torch float32 autograd, no code of this repo — is ~1e-2 away from the float64
    gradient.  Two fp32 implementations with different summation orders can only be compared tightly when they share
    the forward values bit for bit (HIP path vs oracle: they do)."""
    from test_oracle_backward import _step64
    D, H, B = 784, 100, 32
    rng = np.random.default_rng(0)
    lim1, lim2 = np.sqrt(6.0 / (D + 1 + H)), np.sqrt(6.0 / (H + 1 + D))
    p = np.concatenate([(rng.random(H * (D + 1)) * 2 - 1) * lim1 * 3, rng.standard_normal(H) * 0.02,
                        (rng.random(D * (H + 1)) * 2 - 1) * lim2 * 3, rng.standard_normal(D) * 0.02]).astype(np.float32)
    x = rng.random((B, D)).astype(np.float32)

    def grad(dtype):
        pt = torch.tensor(p, dtype=dtype, requires_grad=True)

        def f(u, t):
            W1 = pt[:H * (D + 1)].reshape(D + 1, H).T; b1 = pt[H * (D + 1):H * (D + 1) + H]; o = H * (D + 1) + H
            W2 = pt[o:o + D * (H + 1)].reshape(H + 1, D).T; b2 = pt[o + D * (H + 1):]
            tc = torch.full((u.shape[0], 1), float(t), dtype=dtype)
            return torch.cat([torch.tanh(torch.cat([u, tc], 1) @ W1.T + b1), tc], 1) @ W2.T + b2
        xt = torch.tensor(x, dtype=dtype)
        with torch.no_grad():
            k1 = f(xt, 0.2)
        re, _ = _step64(f, xt, k1, 0.2, 0.1, 1e-3, 1e-3)
        re.backward()
        return pt.grad.numpy().astype(np.float64)

    g64, g32 = grad(torch.float64), grad(torch.float32)
    rel = np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
    assert 1e-4 < rel < 1e-1, rel


def test_biased_index_convention():
    import lrnde_amd  # noqa: F401
    from localregneuralde_jl_amd.layers import biased_index
    assert biased_index(np.float32(0.0), 5) == 0 and biased_index(np.float32(0.999999), 5) == 4
    assert biased_index(np.float32(0.4), 5) == 2 and biased_index(np.float32(0.5), 1) == 0
    # the same integer the C side computes: (int)(r * (float)m), clamped
    for r in np.random.default_rng(1).random(200, dtype=np.float32):
        for m in (1, 2, 7, 33):
            assert biased_index(r, m) == min(int(np.float32(r) * np.float32(m)), m - 1)
