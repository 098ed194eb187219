"""GPU parity of the conv vector field path (lrnde_conv_*) against the oracle (lro_conv_rhs and the
field-agnostic Tsit5 machinery of oracle/lrnde_oracle.c).  Parity here is BY TOLERANCE — rtol 1e-5 of the
output scale for fp32 (BASELINE.json north_star) — because train-mode batch statistics couple all samples
through sums whose order the GPU does not reproduce; accepted/rejected step counts must still be equal."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # of the output scale, fp32 path


def _mods():
    import lrnde_amd as P
    import oracle as O
    return P, O


def _case(W, H, B, seed, act="gelu", train=True, scale=1.0):
    P, O = _mods()
    C, Hc = 8, 64
    rng = np.random.default_rng(seed)
    p = O.glorot_conv_params(C, Hc, seed=seed) * np.float32(scale)
    n1 = 9 * (C + 1) * Hc
    p[n1:n1 + Hc] = rng.uniform(0.5, 1.5, Hc); p[n1 + Hc:n1 + 2 * Hc] = rng.uniform(-0.3, 0.3, Hc)
    n2 = n1 + 2 * Hc + 9 * (Hc + 1) * Hc
    p[n2:n2 + Hc] = rng.uniform(0.5, 1.5, Hc); p[n2 + Hc:n2 + 2 * Hc] = rng.uniform(-0.3, 0.3, Hc)
    p = p.astype(np.float32)
    u = rng.standard_normal((B, C, H, W)).astype(np.float32)
    st = None
    if not train:
        st = np.concatenate([rng.normal(0, 0.2, Hc), rng.uniform(0.5, 2, Hc), rng.normal(0, 0.2, Hc),
                             rng.uniform(0.5, 2, Hc)]).astype(np.float32)
    fld = O.ConvField(W, H, C, Hc, p, act=act, bn_train=train, bn_state=st, nthreads=8)
    h = P.ConvHandle(W, H, C, Hc, act=act, bn_train=train)
    if st is not None:
        h.set_bn_state(st)
    h.set_params(p)
    return fld, h, p, u


def _close(got, ref, rtol=RTOL):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = np.asarray(ref).reshape(got.shape)
    sc = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max())
    assert err <= rtol * sc, f"max abs err {err:.3e} > {rtol:g} * scale {sc:.3e}"


@pytest.mark.parametrize("W,H,B,train,act", [
    (8, 8, 3, True, "gelu"),      # 2 strips of 4 rows? (TR=8: one strip of 64 px)
    (16, 16, 2, True, "gelu"),    # 128-pixel strips, 2 per image
    (28, 28, 2, False, "gelu"),   # MNIST-conv variant: 7 M tiles, test-mode BN
    (32, 32, 2, True, "gelu"),    # CIFAR block shape
    (12, 8, 5, True, "tanh"),     # ragged: 96-pixel strips, odd batch
    (32, 32, 1, False, "identity"),
])
def test_conv_rhs_matches_oracle(W, H, B, train, act):
    fld, h, p, u = _case(W, H, B, seed=W + B, act=act, train=train)
    for t in (0.0, 0.613):
        got = h.rhs(torch.from_numpy(u).cuda(), t)
        _close(got, fld.rhs(u.reshape(B, -1), t))


def test_conv_t_plane_border_classes():
    """only the t channel of conv3 non-zero: du = t * (number of in-image taps)"""
    P, O = _mods()
    W = H = 8; C = 8; Hc = 64
    p = np.zeros(O.lib().lro_conv_param_count(C, Hc), np.float32)
    n3 = 9 * (C + 1) * Hc + 2 * Hc + 9 * (Hc + 1) * Hc + 2 * Hc
    p[n3:].reshape(C, Hc + 1, 3, 3)[:, Hc, :, :] = 1.0
    h = P.ConvHandle(W, H, C, Hc, bn_train=False)
    h.set_params(p)
    du = h.rhs(torch.zeros(1, C, H, W, device="cuda"), 2.0).cpu().numpy()[0]
    cnt = np.full((H, W), 9.0); cnt[0, :] = 6; cnt[-1, :] = 6; cnt[:, 0] = 6; cnt[:, -1] = 6
    cnt[0, 0] = cnt[0, -1] = cnt[-1, 0] = cnt[-1, -1] = 4
    for c in range(C):
        assert np.array_equal(du[c], 2.0 * cnt)


def test_conv_init_dt_and_step_match_oracle():
    P, O = _mods()
    fld, h, p, u = _case(16, 16, 3, seed=11, scale=2.0)
    B = 3
    ud = torch.from_numpy(u).cuda()
    dt_o, k1_o = O.init_dt(fld, u.reshape(B, -1), 0.1, 1.0, 1e-4, 1e-4)
    dt_g, k1_g = h.init_dt(ud, 0.1, 1.0, 1e-4, 1e-4)
    assert abs(float(dt_g) - float(dt_o)) <= 1e-4 * float(dt_o)
    _close(k1_g, k1_o)
    # a step large enough that the embedded error is truncation error, not fp32 cancellation noise
    dt = 0.3
    so = O.tsit5_step(fld, u.reshape(B, -1), k1_o, 0.1, dt, 1e-4, 1e-4)
    sg = h.perform_step(ud, k1_g, 0.1, dt, 1e-4, 1e-4)
    _close(sg["u"], so["u"]); _close(sg["k7"], so["k7"], rtol=5e-5)
    assert float(so["eest"]) > 1e-2
    for key in ("eest", "reg_error", "reg_stiff"):
        assert abs(float(sg[key]) - float(so[key])) <= 5e-3 * abs(float(so[key])) + 1e-12, key


@pytest.mark.parametrize("tol,scale", [(1e-3, 3.0), (1e-4, 1.5)])
def test_conv_solve_matches_oracle_step_for_step(tol, scale):
    P, O = _mods()
    W = H = 16; B = 2
    fld, h, p, u = _case(W, H, B, seed=5, scale=scale)
    ro = O.solve(fld, u.reshape(B, -1), 0.0, 1.0, tol, tol, saveat=[0.37, 1.0])
    rg = h.solve(torch.from_numpy(u).cuda(), 0.0, 1.0, tol, tol, saveat=[0.37, 1.0], trace=True)
    so, sg = ro["stats"], rg["stats"]
    assert (sg["naccept"], sg["nreject"], sg["nf"]) == (so["naccept"], so["nreject"], so["nf"])
    assert so["naccept"] >= 3
    # dt follows EEst^(-1/5-ish); EEst is a cancellation-dominated fp32 quantity (a few % of rounding noise at 1e-4)
    np.testing.assert_allclose(rg["trace"]["dt"], ro["trace"]["dt"][:len(rg["trace"])], rtol=1e-2)
    assert list(rg["t"]) == [np.float32(0.37), np.float32(1.0)]
    for i in range(2):
        _close(rg["u"][i], ro["u"][i], rtol=2e-5)


@pytest.mark.parametrize("mode,reg_type", [("unbiased", "error_estimate"), ("biased", "stiffness_estimate"),
                                           ("none", "error_estimate")])
def test_conv_node_forward_matches_oracle(mode, reg_type):
    P, O = _mods()
    W = H = 8; B = 4
    fld, h, p, u = _case(W, H, B, seed=21, scale=2.0)
    ro = O.node_forward(fld, u.reshape(B, -1), 0.0, 1.0, 1e-3, 1e-3, mode=mode, reg_type=reg_type, t1_or_rand=0.41)
    rg = h.node_forward(torch.from_numpy(u).cuda(), 0.0, 1.0, 1e-3, 1e-3, mode=mode, reg_type=reg_type,
                        t1_or_rand=0.41)
    assert rg["nfe"] == ro["nfe"]
    assert rg["stats"]["naccept"] == ro["stats"]["naccept"] and rg["stats"]["nreject"] == ro["stats"]["nreject"]
    _close(rg["u_end"], ro["u_end"], rtol=2e-5)
    if mode == "none":
        assert rg["reg_val"] == 0.0 and ro["reg_val"] == 0.0
    else:
        # the local step runs at the automatic initial dt, where utilde = dt*sum(btilde_j k_j) (sum btilde = 0)
        # is a difference of nearly equal fp32 numbers: its rms carries several % of rounding noise
        # (f itself agrees to 1e-5 of its scale, test_conv_rhs_matches_oracle)
        assert abs(float(rg["reg_val"]) - float(ro["reg_val"])) <= 1e-1 * abs(float(ro["reg_val"]))
        if mode == "unbiased":
            assert rg["t1"] == ro["t1"] == np.float32(0.41)
        else:  # an accepted step time: equal up to the dt differences
            assert abs(float(rg["t1"]) - float(ro["t1"])) <= 2e-3


def test_conv_layer_surface_runs_the_cifar_block():
    """NeuralODE over the reference's node_core spec (experiments/src/construct.jl:213-218)"""
    P, O = _mods()
    core = P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")),
                             P.Chain(P.Conv((3, 3), 65, 64), P.BatchNorm(64, "gelu")),
                             P.Conv((3, 3), 65, 8)))
    node = P.NeuralODE(core, regularize="unbiased", abstol=1e-3, reltol=1e-3, save_start=False, maxiters=1000)
    ps = torch.from_numpy(P.glorot_conv_params(8, 64, seed=0)).cuda()
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 8, 16, 16)).astype(np.float32)).cuda()
    st = node.initialstates(np.random.default_rng(0))
    sol, st2 = node(x, ps, st)
    assert st2["nfe"] == sol.destats.nf + 9 and st2["reg_val"] > 0 and len(sol.u) == 2
    assert torch.isfinite(sol.u[-1]).all() and tuple(sol.u[-1].shape) == tuple(x.shape)
    bn = st2["model"]["bn_state"]  # running statistics moved away from (0, 1) and are threaded through st.model
    assert bn.shape == (256,) and not torch.allclose(bn[64:128], torch.ones(64, device=bn.device))
    sol_b, st2b = node(x, ps, st2)
    assert not torch.equal(st2b["model"]["bn_state"], bn)
    st_test = dict(st2, training=False)  # Lux.testmode: running statistics are used and left alone
    sol_t, st3 = node(x, ps, st_test)
    assert torch.equal(st3["model"]["bn_state"], bn)
    h = node.handle()
    ref = O.ConvField(16, 16, 8, 64, ps.cpu().numpy(), bn_train=False, bn_state=bn.cpu().numpy(), nthreads=8)
    _close(h.rhs(x, 0.2), ref.rhs(x.cpu().numpy().reshape(2, -1), 0.2))
    assert st3["reg_val"] == 0.0 and st3["nfe"] == sol_t.destats.nf


# ---- bf16 compute mode (BASELINE.json config 4: CIFAR10 block, bf16 MFMA, fp32 accumulate / state / norms) ----
# Tolerances are bf16-appropriate and stated against the OUTPUT SCALE: against the oracle's bf16 emulation
# (same roundings, different summation order: a value near a bf16 rounding boundary can flip, 2^-9 relative on
# that activation) 5e-3; against the plain fp32 oracle 3e-2.
def _bf16_case(W, H, B, seed, train=True):
    P, O = _mods()
    fld32, h32, p, u = _case(W, H, B, seed=seed, train=train)
    st = fld32.bn_state
    fldbf = O.ConvField(W, H, 8, 64, p, act="gelu", bn_train=train, bn_state=st, nthreads=8, bf16=True)
    hbf = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=train, compute_dtype="bf16")
    if st is not None:
        hbf.set_bn_state(st)
    hbf.set_params(p)
    return fld32, fldbf, hbf, u


@pytest.mark.parametrize("W,H,B,train", [(16, 16, 2, True), (32, 32, 2, True), (28, 28, 2, False), (12, 8, 3, True)])
def test_conv_bf16_rhs(W, H, B, train):
    fld32, fldbf, hbf, u = _bf16_case(W, H, B, seed=W + 7, train=train)
    got = hbf.rhs(torch.from_numpy(u).cuda(), 0.3)
    _close(got, fldbf.rhs(u.reshape(B, -1), 0.3), rtol=5e-3)
    _close(got, fld32.rhs(u.reshape(B, -1), 0.3), rtol=3e-2)


def test_conv_bf16_node_forward():
    P, O = _mods()
    W = H = 16; B = 2
    fld32, fldbf, hbf, u = _bf16_case(W, H, B, seed=9)
    ro = O.node_forward(fldbf, u.reshape(B, -1), 0.0, 1.0, 1e-2, 1e-2, mode="unbiased", t1_or_rand=0.41)
    rg = hbf.node_forward(torch.from_numpy(u).cuda(), 0.0, 1.0, 1e-2, 1e-2, mode="unbiased", t1_or_rand=0.41)
    assert rg["stats"]["retcode"] == 0 and abs(rg["stats"]["naccept"] - ro["stats"]["naccept"]) <= 1
    _close(rg["u_end"], ro["u_end"], rtol=2e-2)
    assert rg["reg_val"] > 0


# ---- backward building block: Zygote.pullback(dudt, y, p, t) of the conv field ----
@pytest.mark.parametrize("W,H,B,train,act", [(8, 8, 3, True, "gelu"), (16, 16, 2, True, "gelu"), (32, 32, 2, True, "gelu"),
                                             (28, 28, 2, False, "gelu"), (12, 8, 3, True, "tanh")])
def test_conv_vjp_matches_oracle(W, H, B, train, act):
    """rtol 5e-5 of each output's scale (fp32 MFMA sums vs the oracle's fp64 accumulations)"""
    P, O = _mods()
    fld, h, p, u = _case(W, H, B, seed=W + B + 3, act=act, train=train)
    lam = np.random.default_rng(17).standard_normal(u.shape).astype(np.float32)
    dy_ref, gp_ref = O.conv_vjp(fld, u.reshape(B, -1), 0.41, lam.reshape(B, -1))
    dy, gp = h.vjp(torch.from_numpy(u).cuda(), 0.41, torch.from_numpy(lam).cuda())
    _close(dy, dy_ref, rtol=5e-5)
    gp = gp.cpu().numpy()
    C, Hc = 8, 64
    n1 = 9 * (C + 1) * Hc; n2 = n1 + 2 * Hc; n3 = n2 + 9 * (Hc + 1) * Hc; n4 = n3 + 2 * Hc
    for name, sl in (("w1", slice(0, n1)), ("bn1", slice(n1, n2)), ("w2", slice(n2, n3)), ("bn2", slice(n3, n4)),
                     ("w3", slice(n4, None))):
        sc = np.abs(gp_ref[sl]).max()
        err = np.abs(gp[sl] - gp_ref[sl]).max()
        assert err <= 5e-5 * sc, f"{name}: {err:.3e} vs scale {sc:.3e}"


@pytest.mark.parametrize("reg_type", ["error_estimate", "stiffness_estimate"])
def test_conv_step_reg_grad_matches_oracle(reg_type):
    """d reg_val / d p through one Tsit5 step of the conv field; 2e-3 of the gradient's scale (the seed is a
    cancellation-prone fp32 quantity, see test_conv_node_forward_matches_oracle)"""
    P, O = _mods()
    W = H = 8; B = 3
    fld, h, p, u = _case(W, H, B, seed=31, scale=2.0)
    ud = torch.from_numpy(u).cuda()
    k1 = fld.rhs(u.reshape(B, -1), 0.2)
    dt = 0.25
    g_ref, rv_ref = O.step_reg_grad(fld, u.reshape(B, -1), k1, 0.2, dt, 1e-4, 1e-4, reg_type=reg_type)
    g, rv = h.step_reg_grad(ud, torch.from_numpy(k1.reshape(u.shape)).cuda(), 0.2, dt, 1e-4, 1e-4, reg_type=reg_type)
    assert abs(float(rv) - float(rv_ref)) <= 5e-3 * abs(float(rv_ref))
    g = g.cpu().numpy()
    assert np.abs(g - g_ref).max() <= 1e-2 * np.abs(g_ref).max()
    assert _rel(g, g_ref) <= 1e-2


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)


@pytest.mark.parametrize("mode,w_reg", [("unbiased", 2.5), ("none", 0.0)])
def test_conv_node_backward_matches_oracle(mode, w_reg):
    P, O = _mods()
    W = H = 8; B = 3
    fld, h, p, u = _case(W, H, B, seed=41, scale=1.5)
    wv = np.random.default_rng(3).standard_normal(u.shape).astype(np.float32)
    tol = 1e-4
    bo = O.node_backward(fld, u.reshape(B, -1), 0.0, 1.0, tol, tol, wv.reshape(B, -1), mode=mode, t1_or_rand=0.41, w_reg=w_reg,
                         maxiters=5000)
    bg = h.node_backward(torch.from_numpy(u).cuda(), 0.0, 1.0, tol, tol, torch.from_numpy(wv).cuda(), mode=mode, t1_or_rand=0.41,
                         w_reg=w_reg, maxiters=5000)
    assert bo["retcode"] == 0
    assert bg["stats_fwd"]["naccept"] == bo["stats_fwd"]["naccept"]
    assert abs(bg["stats_bwd"]["naccept"] - bo["stats_bwd"]["naccept"]) <= 1
    assert _rel(bg["dx"].cpu().numpy().reshape(B, -1), bo["dx"]) <= 2e-3
    assert _rel(bg["dp"].cpu().numpy(), bo["dp"]) <= 5e-3


def test_conv_recorded_backward_equals_the_one_call_form():
    """lrnde_conv_node_forward_record + lrnde_conv_node_backward_recorded (one forward per training step) give the
    bits of lrnde_conv_node_backward; a plain solve in between invalidates the record (BADARG, not stale data)."""
    P, O = _mods()
    W = H = 8; B = 3
    fld, h, p, u = _case(W, H, B, seed=45, scale=1.5)
    ud = torch.from_numpy(u).cuda()
    wv = torch.from_numpy(np.random.default_rng(6).standard_normal(u.shape).astype(np.float32)).cuda()
    tol = 1e-3
    st0 = h.get_bn_state().clone()
    one = h.node_backward(ud, 0.0, 1.0, tol, tol, wv, mode="unbiased", t1_or_rand=0.37, w_reg=2.5, maxiters=5000)
    h.set_bn_state(st0)
    fw_plain = h.node_forward(ud, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.37, maxiters=5000)
    h.set_bn_state(st0)
    fw = h.node_forward_record(ud, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.37, maxiters=5000)
    assert torch.equal(fw["u_end"], fw_plain["u_end"]) and fw["reg_val"] == fw_plain["reg_val"] and fw["nfe"] == fw_plain["nfe"]
    two = h.node_backward_recorded(B, wv, w_reg=2.5)
    assert torch.equal(two["dx"], one["dx"]) and torch.equal(two["dp"], one["dp"])
    assert two["stats_bwd"] == one["stats_bwd"]
    h.solve(ud, 0.0, 1.0, tol, tol, saveat=[1.0])
    with pytest.raises(P.LrndeError):
        h.node_backward_recorded(B, wv, w_reg=2.5)


def test_conv_bf16_handle_backward_is_the_fp32_adjoint():
    """compute_dtype=bf16: the forward solve runs the bf16 kernels, derivatives are taken in fp32 (fp32 recompute of the
    hidden activations + the fp32 backward kernels).  So the VJP of a bf16 handle IS the fp32 handle's VJP (same
    kernels, same packs), and node_backward differs from the fp32 handle's only through the bf16 forward trajectory."""
    P, O = _mods()
    W = H = 8; B = 3
    fld, h32, p, u = _case(W, H, B, seed=43, scale=1.5)
    hb = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True, compute_dtype="bf16")
    hb.set_params(p)
    ud = torch.from_numpy(u).cuda()
    lam = torch.from_numpy(np.random.default_rng(4).standard_normal(u.shape).astype(np.float32)).cuda()
    dy32, gp32 = h32.vjp(ud, 0.3, lam)
    hb.rhs(ud, 0.3)  # leaves bf16 activations in the workspace: the VJP must not reuse them
    dyb, gpb = hb.vjp(ud, 0.3, lam)
    assert torch.equal(dyb, dy32) and torch.equal(gpb, gp32)
    # a bf16 f-eval still runs the bf16 kernels afterwards
    _close(hb.rhs(ud, 0.3), fld.rhs(u.reshape(B, -1), 0.3).reshape(u.shape), rtol=3e-2)
    assert not torch.equal(hb.rhs(ud, 0.3), h32.rhs(ud, 0.3))
    wv = torch.from_numpy(np.random.default_rng(5).standard_normal(u.shape).astype(np.float32)).cuda()
    tol = 1e-2  # above the bf16 field's rounding noise, so both solves take comparable steps
    b32 = h32.node_backward(ud, 0.0, 1.0, tol, tol, wv, mode="unbiased", t1_or_rand=0.41, w_reg=2.5, maxiters=5000)
    bb = hb.node_backward(ud, 0.0, 1.0, tol, tol, wv, mode="unbiased", t1_or_rand=0.41, w_reg=2.5, maxiters=5000)
    assert _rel(bb["dx"].cpu().numpy(), b32["dx"].cpu().numpy()) <= 5e-2
    assert _rel(bb["dp"].cpu().numpy(), b32["dp"].cpu().numpy()) <= 5e-2


def test_conv_golden_fixture_on_gpu():
    """the committed fixture tests/golden/conv_block_8x8_b2.npz (generated from the oracle) through the C ABI"""
    P, O = _mods()
    g = np.load(os.path.join(ROOT, "tests", "golden", "conv_block_8x8_b2.npz"))
    h = P.ConvHandle(8, 8, 8, 64, act="gelu", bn_train=True)
    h.set_params(g["params"])
    x = torch.from_numpy(g["x"].reshape(2, 8, 8, 8)).cuda()
    _close(h.rhs(x, float(g["t"])), g["du"])
    dy, gp = h.vjp(x, float(g["t"]), torch.from_numpy(g["lam"].reshape(2, 8, 8, 8)).cuda())
    _close(dy, g["vjp_dy"], rtol=5e-5)
    assert _rel(gp.cpu().numpy(), g["vjp_gp"]) <= 5e-5
    nd = h.node_forward(x, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.41)
    assert nd["nfe"] == int(g["node_nfe"]) and nd["stats"]["naccept"] == int(g["node_naccept"])
    _close(nd["u_end"], g["node_u_end"], rtol=2e-5)


def test_conv_running_statistics_match_oracle():
    """st.model of the layer: every training-mode f-eval advances BatchNorm's running statistics
    (lrnde_conv_get_bn_state); node_forward returns them as they were when the solve returned."""
    P, O = _mods()
    W = H = 8; B = 3
    fld, h, p, u = _case(W, H, B, seed=51, scale=1.5)
    ud = torch.from_numpy(u).cuda()
    st0 = h.get_bn_state().cpu().numpy()
    assert np.array_equal(st0, np.concatenate([np.zeros(64), np.ones(64), np.zeros(64), np.ones(64)]).astype(np.float32))
    h.rhs(ud, 0.2); fld.rhs(u.reshape(B, -1), 0.2)
    np.testing.assert_allclose(h.get_bn_state().cpu().numpy(), fld.bn_run, rtol=2e-5, atol=1e-6)
    rg = h.solve(ud, 0.0, 1.0, 1e-3, 1e-3, saveat=[1.0]); ro = O.solve(fld, u.reshape(B, -1), 0.0, 1.0, 1e-3, 1e-3, saveat=[1.0])
    assert rg["stats"]["nf"] == ro["stats"]["nf"]
    after_solve = h.get_bn_state().cpu().numpy()
    # the stage states of the two solves differ by their dt sequences (1e-3 relative, see above)
    np.testing.assert_allclose(after_solve, fld.bn_run, rtol=5e-3, atol=5e-4)
    # VJP: no update; node_forward: only its solve's f-evals count
    h.vjp(ud, 0.3, ud)
    assert np.array_equal(h.get_bn_state().cpu().numpy(), after_solve)
    h2 = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True); h2.set_params(p)
    h3 = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True); h3.set_params(p)
    h2.node_forward(ud, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.41)
    h3.solve(ud, 0.0, 1.0, 1e-3, 1e-3, saveat=[0.41, 1.0])
    assert np.array_equal(h2.get_bn_state().cpu().numpy(), h3.get_bn_state().cpu().numpy())


# ---- layers around the CIFAR10 NeuralODE ----
@pytest.mark.parametrize("W,H,B,train", [(8, 8, 5, True), (32, 32, 3, True), (16, 16, 4, False)])
def test_cifar_stem_matches_oracle(W, H, B, train):
    P, O = _mods()
    rng = np.random.default_rng(W + B)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    ps = (rng.standard_normal(156) * 0.3).astype(np.float32); ps[140:148] = rng.uniform(0.5, 1.5, 8)
    st = None if train else np.concatenate([rng.normal(0, 0.2, 8), rng.uniform(0.5, 2, 8)]).astype(np.float32)
    g = rng.standard_normal((B, 8, H, W)).astype(np.float32)
    h = P.ConvHandle(W, H, 8, 64, bn_train=train)
    xd, pd = torch.from_numpy(x).cuda(), torch.from_numpy(ps).cuda()
    std = None if st is None else torch.from_numpy(st).cuda()
    _close(h.cifar_stem_forward(xd, pd, std), O.cifar_stem_forward(x, ps, bn_train=train, bn_state=st), rtol=1e-5)
    # the BatchNorm(8) state after the call: advanced from (0, 1) and from a given state in training mode, passed through otherwise
    for s_in in ((None, st) if not train else (None, np.concatenate([rng.normal(0, 0.2, 8), rng.uniform(0.5, 2, 8)]).astype(np.float32))):
        sd = None if s_in is None else torch.from_numpy(s_in).cuda()
        u_g, st_g = h.cifar_stem_forward(xd, pd, sd, return_state=True)
        u_o, st_o = O.cifar_stem_forward(x, ps, bn_train=train, bn_state=s_in, return_state=True)
        _close(u_g, u_o, rtol=1e-5)
        np.testing.assert_allclose(st_g.cpu().numpy(), st_o, rtol=1e-5, atol=1e-6)
    got = h.cifar_stem_backward(xd, pd, torch.from_numpy(g).cuda(), std).cpu().numpy()
    ref = O.cifar_stem_backward(x, ps, g, bn_train=train, bn_state=st)
    assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max()


@pytest.mark.parametrize("W,H,B", [(8, 8, 6), (32, 32, 4)])
def test_cifar_head_matches_oracle(W, H, B):
    P, O = _mods()
    rng = np.random.default_rng(W)
    K = 10
    u = rng.standard_normal((B, 8, H, W)).astype(np.float32)
    ph = (rng.standard_normal(73 + K * H * W + K) * 0.1).astype(np.float32)
    lab = rng.integers(0, K, B).astype(np.int32)
    lo, lg, du, dph = O.cifar_head_ce(u, ph, K, lab)
    h = P.ConvHandle(W, H, 8, 64)
    r = h.cifar_head_ce(torch.from_numpy(u).cuda(), torch.from_numpy(ph).cuda(), K, torch.from_numpy(lab).cuda())
    assert abs(float(r["loss"]) - float(lo)) <= 2e-5 * max(1.0, abs(float(lo)))
    _close(r["logits"], lg, rtol=2e-5)
    _close(r["du"], du, rtol=5e-5)
    got = r["dph"].cpu().numpy()
    assert np.abs(got - dph).max() <= 5e-5 * np.abs(dph).max()


def test_cifar_training_step_runs_and_is_consistent():
    """run_cifar_training_step: loss = CE + w_reg*reg_val, gradients for all three parameter groups; the NeuralODE part
    equals lrnde_conv_node_backward with the head's cotangent (checked against the oracle on the same inputs)"""
    P, O = _mods()
    W = H = 8; B = 4; K = 10
    rng = np.random.default_rng(12)
    core = P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")),
                             P.Chain(P.Conv((3, 3), 65, 64), P.BatchNorm(64, "gelu")), P.Conv((3, 3), 65, 8)))
    node = P.NeuralODE(core, regularize="unbiased", abstol=1e-3, reltol=1e-3, save_start=False, maxiters=2000)
    pn = P.glorot_conv_params(8, 64, seed=0)
    ps = (rng.standard_normal(156) * 0.3).astype(np.float32); ps[140:148] = 1.0; ps[148:156] = 0.0
    ph = (rng.standard_normal(73 + K * H * W + K) * 0.1).astype(np.float32)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    lab = rng.integers(0, K, B).astype(np.int32)
    params = dict(stem=torch.from_numpy(ps).cuda(), neural_ode=torch.from_numpy(pn).cuda(), head=torch.from_numpy(ph).cuda())
    st = node.initialstates(np.random.default_rng(0))
    loss, st_, stats, grads, times = P.run_cifar_training_step(node, params, st, torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), 2.5)
    assert np.isfinite(loss) and times["fwd_time"] > 0 and times["bwd_time"] > 0
    assert grads["stem"].shape == (156,) and grads["neural_ode"].shape == (47560,) and grads["head"].shape == ph.shape
    assert all(torch.isfinite(g).all() and g.abs().max() > 0 for g in grads.values())
    # oracle chain on the same inputs
    import copy
    t1 = np.float32(copy.deepcopy(st["rng"]).random(dtype=np.float32))
    u0 = O.cifar_stem_forward(x, ps)
    fld = O.ConvField(W, H, 8, 64, pn, nthreads=8)
    fo = O.node_forward(fld, u0.reshape(B, -1), 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=t1, maxiters=2000)
    lo, lg, du, dph = O.cifar_head_ce(fo["u_end"].reshape(B, 8, H, W), ph, K, lab)
    assert abs(float(stats["ce_loss"]) - float(lo)) <= 1e-4 * abs(float(lo))
    bo = O.node_backward(fld, u0.reshape(B, -1), 0.0, 1.0, 1e-3, 1e-3, du.reshape(B, -1), mode="unbiased", t1_or_rand=t1, w_reg=2.5,
                         maxiters=2000)
    dstem = O.cifar_stem_backward(x, ps, bo["dx"].reshape(B, 8, H, W))
    assert _rel(grads["neural_ode"].cpu().numpy(), bo["dp"]) <= 1e-2
    assert _rel(grads["head"].cpu().numpy(), dph) <= 1e-3
    assert _rel(grads["stem"].cpu().numpy(), dstem) <= 1e-2


# ---- LRNDE_F32_SPLIT: fp32 results for conv2/conv3 on the fp16 MFMA pipe (hi + lo operand pairs) ----
def _split_case(W, H, B, seed, train=True, scale=1.0):
    P, O = _mods()
    fld, _h32, p, u = _case(W, H, B, seed=seed, train=train, scale=scale)
    h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=train, compute_dtype="f32_split")
    if fld.bn_state is not None:
        h.set_bn_state(fld.bn_state)
    h.set_params(p)
    return fld, h, p, u


@pytest.mark.parametrize("W,H,B,train", [(8, 8, 3, True), (16, 16, 2, True), (28, 28, 2, False), (32, 32, 2, True), (12, 8, 5, True)])
def test_conv_f32_split_rhs_meets_the_fp32_bar(W, H, B, train):
    """the same rtol = 1e-5 of the output scale as the native fp32 path"""
    fld, h, p, u = _split_case(W, H, B, seed=W + B, train=train)
    for t in (0.0, 0.613):
        _close(h.rhs(torch.from_numpy(u).cuda(), t), fld.rhs(u.reshape(B, -1), t))


def test_conv_f32_split_solve_and_backward():
    """equal accepted / rejected step counts; dt follows the oracle's within 10 % (its rounding errors are less correlated
    between the RK stages than the native path's, so a small EEst carries more noise); the backward pass (fp32 transposed
    convs on the split forward's activations) keeps its tolerance"""
    P, O = _mods()
    W = H = 16; B = 2
    fld, h, p, u = _split_case(W, H, B, seed=5, scale=1.5)
    ro = O.solve(fld, u.reshape(B, -1), 0.0, 1.0, 1e-4, 1e-4, saveat=[0.37, 1.0])
    rg = h.solve(torch.from_numpy(u).cuda(), 0.0, 1.0, 1e-4, 1e-4, saveat=[0.37, 1.0], trace=True)
    so, sg = ro["stats"], rg["stats"]
    assert (sg["naccept"], sg["nreject"], sg["nf"]) == (so["naccept"], so["nreject"], so["nf"])
    np.testing.assert_allclose(rg["trace"]["dt"], ro["trace"]["dt"][:len(rg["trace"])], rtol=1e-1)
    for i in range(2):
        _close(rg["u"][i], ro["u"][i], rtol=2e-5)
    lam = np.random.default_rng(17).standard_normal(u.shape).astype(np.float32)
    dy_ref, gp_ref = O.conv_vjp(fld, u.reshape(B, -1), 0.41, lam.reshape(B, -1))
    dy, gp = h.vjp(torch.from_numpy(u).cuda(), 0.41, torch.from_numpy(lam).cuda())
    _close(dy, dy_ref, rtol=5e-5)
    assert _rel(gp.cpu().numpy(), gp_ref) <= 5e-5


@pytest.mark.parametrize("dtype", ["f32", "f32_split", "bf16"])
def test_conv_feval_and_vjp_are_run_to_run_deterministic(dtype):
    """fixed-order reductions everywhere (batch statistics, weight gradients, norms): repeated calls give the same bits"""
    P, O = _mods()
    W = H = 16; B = 3
    h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True, compute_dtype=dtype)
    h.set_params(P.glorot_conv_params(8, 64, seed=1))
    u = torch.from_numpy(np.random.default_rng(0).standard_normal((B, 8, H, W)).astype(np.float32)).cuda()
    a = h.rhs(u, 0.3); b = h.rhs(u, 0.3)
    assert torch.equal(a, b)
    r1 = h.node_forward(u, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.4)
    r2 = h.node_forward(u, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.4)
    assert torch.equal(r1["u_end"], r2["u_end"]) and r1["reg_val"] == r2["reg_val"] and r1["nfe"] == r2["nfe"]
    if dtype != "bf16":
        lam = torch.ones_like(u)
        d1, g1 = h.vjp(u, 0.3, lam); d2, g2 = h.vjp(u, 0.3, lam)
        assert torch.equal(d1, d2) and torch.equal(g1, g2)


# ---- size-independent properties at BASELINE's full conv shapes (the oracle needs minutes there) ----
def test_conv_full_size_properties():
    """CIFAR10 block 32x32x8, B=256 (BASELINE config 4) and the 28x28 B=512 field (config 2-ii):
    (1) with running statistics the field is per-sample: f(u)[rows] == f(u[rows]) bit for bit, whatever else is in the batch;
    (2) the VJP is linear in the cotangent and satisfies <w, J v> = <J^T w, v> against a central difference of the field;
    (3) with batch statistics a permutation of the batch permutes the output (to the summation-order tolerance)."""
    P, O = _mods()
    rng = np.random.default_rng(21)
    for (W, H, B) in ((32, 32, 256), (28, 28, 512)):
        p = P.glorot_conv_params(8, 64, seed=1)
        u = torch.from_numpy(rng.standard_normal((B, 8, H, W)).astype(np.float32)).cuda()
        he = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=False)
        he.set_params(p)
        full = he.rhs(u, 0.3)
        rows = torch.tensor([0, 3, B // 2, B - 1], device="cuda")
        assert torch.equal(full[rows], he.rhs(u[rows].contiguous(), 0.3))
        ht = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True)
        ht.set_params(p)
        w1 = torch.from_numpy(rng.standard_normal(u.shape).astype(np.float32)).cuda()
        w2 = torch.from_numpy(rng.standard_normal(u.shape).astype(np.float32)).cuda()
        d1, g1 = ht.vjp(u, 0.3, w1); d2, g2 = ht.vjp(u, 0.3, w2); d12, g12 = ht.vjp(u, 0.3, w1 + w2)
        assert _rel((d1 + d2).cpu().numpy(), d12.cpu().numpy()) <= 2e-5 and _rel((g1 + g2).cpu().numpy(), g12.cpu().numpy()) <= 2e-5
        v = torch.from_numpy(rng.standard_normal(u.shape).astype(np.float32)).cuda()
        eps = 1e-2
        jv = (ht.rhs(u + eps * v, 0.3).double() - ht.rhs(u - eps * v, 0.3).double()) / (2 * eps)
        lhs = float((w1.double() * jv).sum()); rhs = float((d1.double() * v.double()).sum())
        assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs)), (lhs, rhs)
        perm = torch.from_numpy(rng.permutation(B)).cuda()
        fa = ht.rhs(u, 0.3); fb = ht.rhs(u[perm].contiguous(), 0.3)
        assert _rel(fb.cpu().numpy(), fa[perm].cpu().numpy()) <= 1e-5


def test_conv_bf16_full_size_properties():
    """BASELINE.json config 4 at its own size and dtype: the CIFAR10 block 32x32x8, B=256 with bf16 handles.
    (1) running statistics: the bf16 field is per-sample — f(u)[rows] == f(u[rows]) bit for bit;
    (2) the bf16 f-eval is within 3e-2 of the fp32 handle's f-eval at full size (bf16 operands, 8 mantissa bits; the same
        bound the small-shape oracle tests use), train-mode and test-mode statistics;
    (3) batch statistics: a permutation of the batch permutes the output (bf16 rounding + summation-order tolerance 2e-2);
    (4) determinism: two evaluations and two adaptive forward passes at the experiment's tolerance are bitwise equal, the
        solve finishes (retcode 0) and reports the NFE the bench line quotes for bf16."""
    P, O = _mods()
    rng = np.random.default_rng(22)
    W, H, B = 32, 32, 256
    p = P.glorot_conv_params(8, 64, seed=1)
    u = torch.from_numpy(rng.standard_normal((B, 8, H, W)).astype(np.float32)).cuda()
    for train in (False, True):
        hb = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=train, compute_dtype="bf16")
        hf = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=train)
        hb.set_params(p); hf.set_params(p)
        fb, ff = hb.rhs(u, 0.3), hf.rhs(u, 0.3)
        assert torch.isfinite(fb).all()
        err = float((fb - ff).abs().max() / ff.abs().max())
        assert err <= 3e-2, err
        assert torch.equal(fb, hb.rhs(u, 0.3))
        if not train:
            rows = torch.tensor([0, 5, B // 2, B - 1], device="cuda")
            assert torch.equal(fb[rows], hb.rhs(u[rows].contiguous(), 0.3))
        else:
            perm = torch.from_numpy(rng.permutation(B)).cuda()
            fp = hb.rhs(u[perm].contiguous(), 0.3)
            assert _rel(fp.cpu().numpy(), fb[perm].cpu().numpy()) <= 2e-2
            r1 = hb.node_forward(u, 0.0, 1.0, 1e-4, 1e-4, mode="unbiased", t1_or_rand=0.4, maxiters=10000)
            hb.set_bn_state(np.concatenate([np.zeros(64), np.ones(64), np.zeros(64), np.ones(64)]).astype(np.float32))
            r2 = hb.node_forward(u, 0.0, 1.0, 1e-4, 1e-4, mode="unbiased", t1_or_rand=0.4, maxiters=10000)
            assert torch.equal(r1["u_end"], r2["u_end"]) and r1["nfe"] == r2["nfe"] and r1["reg_val"] == r2["reg_val"]
            assert r1["stats"]["retcode"] == 0 and torch.isfinite(r1["u_end"]).all()
            rf = hf.node_forward(u, 0.0, 1.0, 1e-4, 1e-4, mode="unbiased", t1_or_rand=0.4, maxiters=10000)
            print(f"CIFAR block B=256 tol 1e-4: bf16 nfe {r1['nfe']} (accepted {r1['stats']['naccept']}), fp32 nfe {rf['nfe']}")
            # both integrate the same ODE to the same tolerance: the end states agree to the bf16 field's accuracy
            assert float((r1["u_end"] - rf["u_end"]).abs().max() / rf["u_end"].abs().max()) <= 5e-2


@pytest.mark.parametrize("W,H,B", [(64, 4, 2), (128, 2, 2)])
def test_conv_wide_images(W, H, B):
    """widths whose halo tile exceeds the default 64 KiB of dynamic LDS (the maximum supported width is 128)"""
    P, O = _mods()
    fld, h, p, u = _case(W, H, B, seed=W, scale=1.2)
    ud = torch.from_numpy(u).cuda()
    _close(h.rhs(ud, 0.3), fld.rhs(u.reshape(B, -1), 0.3).reshape(u.shape))
    lam = np.random.default_rng(5).standard_normal(u.shape).astype(np.float32)
    if W <= 124:
        dy, gp = h.vjp(ud, 0.3, torch.from_numpy(lam).cuda())
        dyo, gpo = O.conv_vjp(fld, u.reshape(B, -1), 0.3, lam.reshape(B, -1))
        _close(dy, dyo.reshape(u.shape), rtol=5e-5)
        assert _rel(gp.cpu().numpy(), gpo) <= 5e-5
    else:  # the 64 x 64 weight-gradient kernel's tiles need 166 KB at W = 128: refused, not mis-launched
        with pytest.raises(P.LrndeError, match="weight-gradient"):
            h.vjp(ud, 0.3, torch.from_numpy(lam).cuda())
    hb = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True, compute_dtype="bf16")
    hb.set_params(p)
    _close(hb.rhs(ud, 0.3), fld.rhs(u.reshape(B, -1), 0.3).reshape(u.shape), rtol=3e-2)


import os as _os


@pytest.mark.parametrize("seed", list(range(int(_os.environ.get("LRNDE_SOAK_SEEDS", "6")))))
def test_conv_rhs_and_vjp_soak(seed):
    """the conv field's f-eval and VJP with random image size (W % 4 == 0 up to 32, any height >= 2), batch, activation, BN mode
    and time, fp32: 1e-5 of scale against the oracle (f-eval), 2e-5 (state cotangent), 1e-4 (parameter cotangent, batch sums)"""
    P, O = _mods()
    rng = np.random.default_rng(30_000 + seed)
    W = int(rng.choice([4, 8, 12, 16, 28, 32])); H = int(rng.choice([2, 3, 8, 9, 16, 28, 32]))
    B = int(rng.choice([1, 2, 3, 5])); act = str(rng.choice(["gelu", "tanh", "identity"])); train = bool(rng.integers(0, 2))
    if train and B * W * H < 8: B = 4
    fld, h, p, u = _case(W, H, B, seed=seed, act=act, train=train)
    t = float(np.float32(rng.random()))
    ud = torch.from_numpy(u).cuda()
    _close(h.rhs(ud, t), fld.rhs(u.reshape(B, -1), t))
    lam = rng.standard_normal(u.shape).astype(np.float32)
    dy_o, gp_o = O.conv_vjp(fld, u.reshape(B, -1), t, lam.reshape(B, -1))
    dy_g, gp_g = h.vjp(ud, t, torch.from_numpy(lam).cuda())
    _close(dy_g, dy_o, rtol=2e-5)
    _close(gp_g, gp_o, rtol=1e-4)
