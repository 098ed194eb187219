"""Every diagnostic switch of DESIGN.md 4.6 that selects an alternative product path, run in the GPU suite: the alternative
must give the default path's bits ("each is a path the default GPUTEST run does not cover unless a test
sets it").  The switches are process-wide options set through the hook lrnde_set_option (the environment variable of the
same name gives the initial value), so both paths run in this one process."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEFAULTS = {"LRNDE_NO_QTILE": 0, "LRNDE_QTILE_MAX_B": 2048, "LRNDE_NO_FUSE": 0, "LRNDE_DENSE_COPY": 0, "LRNDE_NO_OVERLAP": 0,
            "LRNDE_NO_SDE_FAST": 0, "LRNDE_SDE_HOST_LOOP": 0, "LRNDE_NO_QVJP": 0, "LRNDE_ADJ_ERR_ONE_LAUNCH": 0, "LRNDE_ADJ_MU_FOLD": 0,
            "LRNDE_ADJ_HOST": 0, "LRNDE_VJP_QCOLS": 4, "LRNDE_ADJ_OVERLAP": 0, "LRNDE_PGRAD_TS": 0, "LRNDE_ADJ_NO_REUSE": 0, "LRNDE_SDE_NO_PERSIST": 0, "LRNDE_SDE_COOP_LAUNCH": 0, "LRNDE_SDE_PERSIST_STALL": 0, "LRNDE_SDE_HOST_INITDT": 0, "LRNDE_SDE_BWD_LDSACC": 0, "LRNDE_SDE_BWD_NO_DEFER": 0, "LRNDE_SDE_BWD_NO_RESIDENT": 0, "LRNDE_SDE_NO_MARCH": 0, "LRNDE_NO_SDE_BWD_FUSED": 0,
            "LRNDE_FEED_T": 3, "LRNDE_FEED_E": 1, "LRNDE_FEED_M": 2}


@pytest.fixture
def options(gpu_pkg):
    yield gpu_pkg.set_option
    for k, v in DEFAULTS.items():
        gpu_pkg.set_option(k, v)


def _layer_pass(P, B=36, mode="unbiased", tol=1e-5):
    """recorded forward + backward of the MNIST-shaped layer on a fresh handle: everything the ODE switches can touch"""
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    D, H = 784, 100
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    p = P.glorot_params(model, seed=3) * np.float32(1.5)
    x = np.random.default_rng(5).random((B, D), dtype=np.float32)
    g = np.random.default_rng(6).standard_normal((B, D)).astype(np.float32)
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(p))
    xd, gd = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    f = h.node_forward(xd, 0.0, 1.0, tol, tol, mode=mode, t1_or_rand=0.41, maxiters=5000)
    fr = h.node_forward_record(xd, 0.0, 1.0, tol, tol, mode=mode, t1_or_rand=0.41, maxiters=5000)
    b = h.node_backward_recorded(gd, w_reg=2.0)
    one = h.node_backward(xd, 0.0, 1.0, tol, tol, gd, mode=mode, t1_or_rand=0.41, w_reg=2.0, maxiters=5000)
    k1 = h.rhs(xd, 0.1)
    st = h.perform_step(xd, k1, 0.1, 0.05, tol, tol)
    out = dict(u_end=f["u_end"].cpu().numpy(), nfe=f["nfe"], reg=f["reg_val"], stats=f["stats"], u_rec=fr["u_end"].cpu().numpy(),
               dx=b["dx"].cpu().numpy(), dp=b["dp"].cpu().numpy(), sb=b["stats_bwd"], dx1=one["dx"].cpu().numpy(),
               dp1=one["dp"].cpu().numpy(), u_step=st["u"].cpu().numpy(), eest=st["eest"])
    h.close() if hasattr(h, "close") else None
    return out


def _same(a, b, what):
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), (what, k, float(np.abs(a[k] - b[k]).max()))
        elif isinstance(a[k], dict):
            for kk in ("nf", "naccept", "nreject", "iters", "dt_init", "t_final"):
                assert a[k][kk] == b[k][kk], (what, k, kk, a[k], b[k])
        else:
            assert a[k] == b[k], (what, k, a[k], b[k])


@pytest.fixture(scope="module")
def baseline(gpu_pkg):
    for k, v in DEFAULTS.items():
        gpu_pkg.set_option(k, v)
    return {m: _layer_pass(gpu_pkg, mode=m) for m in ("unbiased", "biased")}


@pytest.mark.parametrize("switch", [{"LRNDE_NO_QTILE": 1}, {"LRNDE_QTILE_MAX_B": 16}, {"LRNDE_NO_QTILE": 1, "LRNDE_NO_FUSE": 1},
                                    {"LRNDE_DENSE_COPY": 1}, {"LRNDE_NO_OVERLAP": 1}, {"LRNDE_NO_QVJP": 1},
                                    {"LRNDE_ADJ_ERR_ONE_LAUNCH": 1}, {"LRNDE_ADJ_MU_FOLD": 1}, {"LRNDE_ADJ_HOST": 1}, {"LRNDE_VJP_QCOLS": 2}, {"LRNDE_ADJ_OVERLAP": 1}, {"LRNDE_PGRAD_TS": 1}, {"LRNDE_PGRAD_TS": 2}, {"LRNDE_PGRAD_TS": 3}, {"LRNDE_ADJ_NO_REUSE": 1},
                                    {"LRNDE_PGRAD_TS": 2, "LRNDE_ADJ_MU_FOLD": 1},
                                    {"LRNDE_FEED_T": 100, "LRNDE_FEED_E": 1, "LRNDE_FEED_M": 2}, {"LRNDE_FEED_T": 0, "LRNDE_FEED_E": 0, "LRNDE_FEED_M": 1}],
                         ids=lambda d: "+".join(f"{k[6:]}={v}" for k, v in d.items()))
@pytest.mark.parametrize("mode", ["unbiased", "biased"])
def test_ode_switch_gives_the_default_bits(gpu_pkg, options, baseline, switch, mode):
    for k, v in switch.items():
        options(k, v)
    got = _layer_pass(gpu_pkg, mode=mode)
    _same(baseline[mode], got, (switch, mode))


def _sde_pass(P):
    import torch
    from localregneuralde_jl_amd.layers import _mlp_desc
    D, H, B, nfine = 32, 64, 48, 64
    rng = np.random.default_rng(0)
    pd = (rng.standard_normal(H * D + H + D * H + D) * 0.2).astype(np.float32)
    pg = (rng.standard_normal(D * D + D) * 0.1).astype(np.float32)
    u0 = rng.standard_normal((B, D)).astype(np.float32)
    dt = np.float32(1.0 / nfine)
    dW = (rng.standard_normal((nfine, B, D)) * np.sqrt(dt)).astype(np.float32)
    W = np.concatenate([np.zeros((1, B, D), np.float32), np.cumsum(dW, axis=0, dtype=np.float32)])
    h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
    h.set_params(pd, pg)
    ud = torch.from_numpy(u0).cuda()
    s1 = h.euler_heun_step(ud, torch.from_numpy(dW[0]).cuda(), 0.0, float(dt), 0.14, 0.14, 1.0 / 6.0)
    sf = h.solve_fixed(ud, torch.from_numpy(dW[:20]).cuda(), 0.0, float(dt), 0.14, 0.14, 1.0 / 6.0)
    sa = h.solve_adaptive(ud, torch.from_numpy(W).cuda(), 0.0, 1.0, 0.05, 0.05)
    return dict(u1=s1["u"].cpu().numpy(), e1=s1["eest"], r1=s1["reg_val"], uf=sf["u"].cpu().numpy(), ef=sf["eest"], rf=sf["reg_val"],
                ua=sa["u_end"].cpu().numpy(), sa=sa["stats"],
                tr=np.stack([sa["trace"][f].astype(np.float64) for f in ("t", "dt", "eest", "accepted")]))


@pytest.mark.parametrize("switch", [{"LRNDE_NO_SDE_FAST": 1}, {"LRNDE_SDE_HOST_LOOP": 1}, {"LRNDE_SDE_NO_PERSIST": 1}, {"LRNDE_SDE_COOP_LAUNCH": 1}, {"LRNDE_SDE_PERSIST_STALL": 1}, {"LRNDE_SDE_NO_MARCH": 1}], ids=lambda d: "+".join(k[6:] for k in d))
def test_sde_switch_gives_the_default_bits(gpu_pkg, options, switch):
    for k, v in DEFAULTS.items():
        gpu_pkg.set_option(k, v)
    base = _sde_pass(gpu_pkg)
    for k, v in switch.items():
        options(k, v)
    got = _sde_pass(gpu_pkg)
    _same(base, got, switch)


def _sde_layer_pass(P, mode):
    """the NeuralDSDE layer's recorded forward + pullback at the one-launch step's shape, automatic initial dts"""
    import torch
    from localregneuralde_jl_amd.layers import _mlp_desc
    D, H, B, nfine = 32, 64, 48, 64
    rng = np.random.default_rng(3)
    pd = (rng.standard_normal(H * D + H + D * H + D) * 0.25).astype(np.float32)
    pg = (rng.standard_normal(D * D + D) * 0.08).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    hh = np.float32(1.0 / nfine)
    dW = (rng.standard_normal((nfine, B, D)) * np.sqrt(hh)).astype(np.float32)
    W = torch.from_numpy(np.concatenate([np.zeros((1, B, D), np.float32), np.cumsum(dW, axis=0, dtype=np.float32)])).cuda()
    z = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
    h.set_params(pd, pg)
    fw = h.node_forward_record(x, W, 0.0, 1.0, 0.1, 0.1, z_local=z, mode=mode, t1_or_rand=0.37, saveat=(), save_start=-1)
    ns = fw["u"].shape[0]
    du = torch.from_numpy(rng.standard_normal((ns, B, D)).astype(np.float32)).cuda()
    bw = h.node_backward_recorded(du, w_reg=2.0 if mode != "none" else 0.0)
    out = dict(u=fw["u"].cpu().numpy(), t=fw["t"], reg=fw["reg_val"], st=fw["stats"], nf=fw["nfe_drift"], ng=fw["nfe_diffusion"], t1=fw["t1"])
    for k, v in bw.items():
        if hasattr(v, "cpu"):
            out["bw_" + k] = v.cpu().numpy()
    return out


@pytest.mark.parametrize("switch", [{"LRNDE_SDE_HOST_INITDT": 1}, {"LRNDE_SDE_NO_PERSIST": 1}, {"LRNDE_SDE_COOP_LAUNCH": 1}, {"LRNDE_SDE_PERSIST_STALL": 1}, {"LRNDE_SDE_HOST_LOOP": 1}, {"LRNDE_NO_SDE_FAST": 1},
                                    {"LRNDE_SDE_BWD_LDSACC": 1}, {"LRNDE_SDE_BWD_NO_DEFER": 1}, {"LRNDE_SDE_BWD_NO_RESIDENT": 1}],
                         ids=lambda d: "+".join(k[6:] for k in d))
@pytest.mark.parametrize("mode", ["unbiased", "biased", "none"])
def test_sde_layer_switch_gives_the_default_bits(gpu_pkg, options, switch, mode):
    for k, v in DEFAULTS.items():
        gpu_pkg.set_option(k, v)
    base = _sde_layer_pass(gpu_pkg, mode)
    assert base["st"]["naccept"] > 1
    for k, v in switch.items():
        options(k, v)
    got = _sde_layer_pass(gpu_pkg, mode)
    if "LRNDE_NO_SDE_FAST" in switch or "LRNDE_SDE_BWD_LDSACC" in switch or "LRNDE_SDE_BWD_NO_DEFER" in switch or "LRNDE_SDE_BWD_NO_RESIDENT" in switch:   # the other pullbacks add the parameter gradients in another order
        for k in ("bw_dx", "bw_dp_drift", "bw_dp_diff"):
            x, y = base.pop(k), got.pop(k)
            assert np.allclose(x, y, rtol=2e-5, atol=2e-5 * float(np.abs(x).max())), (switch, mode, k)
    _same(base, got, (switch, mode))


def test_unknown_option_is_rejected(gpu_pkg):
    with pytest.raises(gpu_pkg.LrndeError):
        gpu_pkg.set_option("LRNDE_NO_SUCH_SWITCH", 1)
