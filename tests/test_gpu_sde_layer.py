"""SURVEY.md §8 a12 — the NeuralDSDE layer as the reference runs it (src/layers/neural_sde.jl:50-123): ADAPTIVE solve inside the
layer, local step at (sol(t1), t1), user saveat + _CorrectedDESolution, and the pullback through the recorded accepted steps.

* forward: `lrnde_sde_node_forward_record` == the oracle's loop (oracle.sde_node_forward: the C oracle's Euler-Heun step under the
  same controller / interpolant / initial-dt rules, in float32 numpy) BIT FOR BIT — every state of the series, reg_val, the
  closures' call counts, accepted / rejected steps.  BASELINE config 5 (state 32, hidden 64, B = 512, abstol = reltol = 0.14)
  and shapes on every branch of the one-launch kernel's templates (D <= 64, H <= 128, ragged sizes) plus one outside them.
* backward: `lrnde_sde_node_backward_recorded` against float64 torch autograd through a restatement of the same steps on the
  recorded grid (written in test_gpu_sde_gradients.py, no code shared with the library): 5e-6 of each gradient's norm.
* the assertions of test/runtests.jl:340-433 on the reference's own toy model (2 -> 4 -> 2, gelu)."""
import numpy as np
import pytest
import torch

from test_gpu_sde_gradients import _eh_reg64, _eh_step64, _fields64, _params, _rel

pytestmark = pytest.mark.gpu
f32 = np.float32


def _path(rng, nfine, B, D, span=1.0):
    h = f32(span) / f32(nfine)
    inc = (rng.standard_normal((nfine, B, D)) * np.sqrt(h)).astype(f32)
    return np.concatenate([np.zeros((1, B, D), f32), np.cumsum(inc, axis=0, dtype=f32)], axis=0)


def _setup(P, O, D, H, B, nfine, seed, act="tanh", scale=2.0):
    from localregneuralde_jl_amd.layers import _mlp_desc
    pd, pg = _params(D, H, seed)
    pd = (pd * f32(scale)).astype(f32)
    rng = np.random.default_rng(seed + 100)
    x = rng.standard_normal((B, D)).astype(f32)
    W = _path(rng, nfine, B, D)
    z = rng.standard_normal((B, D)).astype(f32)
    h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, act), P.Dense(H, D))))
    h.set_params(pd, pg)
    drift = O.MlpField(D, H, pd, time_dep=False, act=act, nthreads=4)
    p2 = np.concatenate([np.eye(D, dtype=f32).ravel(), np.zeros(D, f32), pg])
    diff = O.MlpField(D, D, p2, time_dep=False, act="identity", nthreads=4)
    return h, drift, diff, pd, pg, x, W, z


def _check_forward(got, ref, what):
    assert got["stats"]["naccept"] == ref["naccept"] and got["stats"]["nreject"] == ref["nreject"], (what, got["stats"], ref["naccept"], ref["nreject"])
    assert got["nfe_drift"] == ref["nfe_drift"] and got["nfe_diffusion"] == ref["nfe_diffusion"], what
    assert np.array_equal(got["t"], ref["t"]), (what, got["t"], ref["t"])
    assert got["reg_val"] == ref["reg_val"], (what, got["reg_val"], ref["reg_val"])
    gu = got["u"].cpu().numpy()
    assert gu.shape == ref["u"].shape, (what, gu.shape, ref["u"].shape)
    assert np.array_equal(gu, ref["u"]), (what, float(np.abs(gu - ref["u"]).max()))
    assert got["t1"] == ref["t1"], what


@pytest.mark.parametrize("D,H,B,tol,nfine", [(32, 64, 512, 0.14, 128),      # BASELINE config 5
                                             (32, 64, 40, 0.02, 256), (2, 4, 1, 0.05, 64), (20, 48, 33, 0.05, 64), (64, 128, 17, 0.1, 64),
                                             (33, 100, 9, 0.1, 32), (8, 40, 130, 0.05, 64), (48, 113, 5, 0.1, 32),
                                             (72, 32, 6, 0.1, 32)])         # D > 64: the generic kernels + the host-controlled loop
@pytest.mark.parametrize("mode", ["unbiased", "biased", "none"])
def test_adaptive_layer_forward_equals_the_oracle_loop(oracle, gpu_pkg, D, H, B, tol, nfine, mode):
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=7)
    xd, Wd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(z).cuda()
    kw = dict(mode=mode, t1_or_rand=0.43, saveat=(), save_start=-1)
    got = h.node_forward_record(xd, Wd, 0.0, 1.0, tol, tol, z_local=zd, **kw)
    ref = oracle.sde_node_forward(drift, diff, x, W, 0.0, 1.0, tol, tol, z_local=z, **kw)
    what = f"D={D} H={H} B={B} tol={tol} nfine={nfine} {mode}"
    _check_forward(got, ref, what)
    assert (got["reg_val"] == 0) == (mode == "none")                 # test/runtests.jl:357, 384, 417
    assert ref["naccept"] >= 2
    print(f"{what}: accepted {ref['naccept']}, rejected {ref['nreject']}, dt0 {ref['dt0']:.4g}, series {len(ref['t'])}, reg_val {ref['reg_val']:.4g}")


@pytest.mark.parametrize("mode,saveat,save_start", [("unbiased", (0.25, 0.5, 1.0), 0), ("unbiased", (0.0, 0.43, 0.7, 1.0), -1),
                                                    ("none", (0.1, 0.9), 1), ("biased", (0.2, 0.4, 0.8, 1.0), 0), ("biased", (), 0),
                                                    ("unbiased", (), 1)])
def test_adaptive_layer_user_saveat_and_corrected_solution(oracle, gpu_pkg, mode, saveat, save_start):
    """the layer's saveat rules (src/layers/neural_ode.jl:102-116): :unbiased adds t1 to a user saveat and `_CorrectedDESolution`
    drops it again (src/utils.jl:31-33) — also when the user's list contains t1 itself (0.43 below: both entries go, as
    `sol.u[t1 .!= sol.t]` does); :biased draws t1 from sol.t[1:end-1]; explicit and default save_start"""
    D, H, B, nfine, tol = 32, 64, 24, 64, 0.05
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=11)
    xd, Wd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(z).cuda()
    kw = dict(mode=mode, t1_or_rand=0.43, saveat=saveat, save_start=save_start)
    got = h.node_forward_record(xd, Wd, 0.0, 1.0, tol, tol, z_local=zd, **kw)
    ref = oracle.sde_node_forward(drift, diff, x, W, 0.0, 1.0, tol, tol, z_local=z, **kw)
    _check_forward(got, ref, str(kw))
    if mode == "unbiased" and saveat:
        assert f32(0.43) not in got["t"] and len(got["t"]) == len([s for s in saveat if f32(s) != f32(0.43)])
    if mode == "biased":
        assert got["t1"] in got["t"][:-1]


def _autograd64(pd, pg, D, H, x, W, ref, du_series, w_reg, tol, delta=1.0 / 6.0):
    """loss = sum_j <du_j, sol.u[j]> + w_reg * reg_val in float64 over the RECORDED grid (ref['steps']), by torch autograd"""
    pdt = torch.tensor(pd, dtype=torch.float64, requires_grad=True)
    pgt = torch.tensor(pg, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    f, g = _fields64(pdt, pgt, D, H)
    nfine = W.shape[0] - 1
    hh = 1.0 / nfine
    Wt = torch.tensor(W, dtype=torch.float64)
    states, u = [], xt
    for (i, m) in ref["steps"]:
        u = _eh_step64(f, g, u, Wt[i + m] - Wt[i], m * hh)[0]
        states.append(u)
    loss = 0.0
    for j, (ts, k, th) in enumerate(ref["series"]):
        if k < 0:
            val = xt
        else:
            a = xt if k == 0 else states[k - 1]
            val = (1.0 - float(th)) * a + float(th) * states[k]
        loss = loss + (val * torch.tensor(du_series[j], dtype=torch.float64)).sum()
    if ref["u1"] is not None and w_reg != 0.0:
        loss = loss + w_reg * _eh_reg64(f, g, torch.tensor(ref["u1"], dtype=torch.float64), torch.tensor(ref["dW_local"], dtype=torch.float64),
                                        float(ref["dt_local"]), tol, tol, delta)
    loss.backward()
    return xt.grad.numpy(), pdt.grad.numpy(), pgt.grad.numpy()


@pytest.mark.parametrize("D,H,B,tol,nfine,mode,saveat", [(32, 64, 64, 0.14, 64, "unbiased", ()), (32, 64, 512, 0.14, 128, "unbiased", ()),
                                                         (32, 64, 16, 0.05, 64, "biased", ()), (20, 48, 9, 0.05, 64, "unbiased", (0.3, 0.77, 1.0)),
                                                         (2, 4, 3, 0.05, 32, "none", (0.5, 1.0)), (72, 32, 6, 0.1, 32, "unbiased", ())])
def test_adaptive_layer_pullback_matches_float64_autograd(oracle, gpu_pkg, D, H, B, tol, nfine, mode, saveat):
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=21, scale=1.5)
    xd, Wd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(z).cuda()
    kw = dict(mode=mode, t1_or_rand=0.37, saveat=saveat, save_start=-1)
    got = h.node_forward_record(xd, Wd, 0.0, 1.0, tol, tol, z_local=zd, **kw)
    ref = oracle.sde_node_forward(drift, diff, x, W, 0.0, 1.0, tol, tol, z_local=z, **kw)
    _check_forward(got, ref, str(kw))
    ns = len(ref["t"])
    du = np.random.default_rng(5).standard_normal((ns, B, D)).astype(f32)
    w_reg = 2.0
    bw = h.node_backward_recorded(torch.from_numpy(du).cuda(), w_reg=w_reg)
    gx, gpd, gpg = _autograd64(pd, pg, D, H, x, W, ref, du, w_reg, tol)
    errs = {n: _rel(a.cpu().numpy(), b) for n, a, b in (("dx", bw["dx"], gx), ("dp_drift", bw["dp_drift"], gpd), ("dp_diff", bw["dp_diff"], gpg))}
    print(f"D={D} H={H} B={B} {mode} saveat={saveat}: {ref['naccept']} recorded steps, series {ns}; rel err vs float64 autograd " +
          ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < 5e-6, (k, v)


@pytest.mark.parametrize("D,H,B,tol,nfine,mode", [(32, 64, 64, 0.14, 64, "unbiased"), (32, 64, 512, 0.14, 128, "biased"), (20, 48, 9, 0.05, 64, "unbiased"),
                                                  (72, 32, 6, 0.1, 32, "unbiased")])
def test_adaptive_layer_regulariser_gradient_alone_matches_float64_autograd(oracle, gpu_pkg, D, H, B, tol, nfine, mode):
    """zero cotangents on the series, w_reg = 1: what comes back is d reg_val / d ps alone (beside the series' cotangents it is a few
    1e-7 of the gradient — a regulariser counted twice would pass the test above), and d reg_val / d x is exactly zero
    (test/runtests.jl:388-392: `=== nothing`)"""
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=33, scale=1.5)
    xd, Wd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(z).cuda()
    # an explicit initial dt: the automatic one of these fields is ~1e-6 and reg_val ~ dt^2.5 would be below 1e-15
    kw = dict(mode=mode, t1_or_rand=0.41, saveat=(), save_start=-1, dt0=0.05)
    got = h.node_forward_record(xd, Wd, 0.0, 1.0, tol, tol, z_local=zd, **kw)
    ref = oracle.sde_node_forward(drift, diff, x, W, 0.0, 1.0, tol, tol, z_local=z, **kw)
    _check_forward(got, ref, str(kw))
    assert float(ref["reg_val"]) > 1e-6
    ns = len(ref["t"])
    du = np.zeros((ns, B, D), f32)
    bw = h.node_backward_recorded(torch.from_numpy(du).cuda(), w_reg=1.0)
    gx, gpd, gpg = _autograd64(pd, pg, D, H, x, W, ref, du, 1.0, tol)
    assert not bw["dx"].cpu().numpy().any()
    assert np.abs(gpd).max() > 0 and np.abs(gpg).max() > 0
    for k, a, b in (("dp_drift", bw["dp_drift"], gpd), ("dp_diff", bw["dp_diff"], gpg)):
        e = _rel(a.cpu().numpy(), b)
        print(f"D={D} H={H} B={B} {mode}: regulariser alone, {k} rel err {e:.2e} (max |ref| {np.abs(b).max():.3e})")
        assert e < 2e-5, (k, e)


@pytest.mark.parametrize("regularize", ["none", "unbiased", "biased"])
def test_reference_testitems_on_the_toy_model(gpu_pkg, regularize):
    """test/runtests.jl:340-433 — NeuralDSDE(Chain(Dense(2 => 4, gelu), Dense(4 => 2)), Dense(2 => 2); regularize, tspan = (0, 1)),
    x = randn(2, 1): output finite Float32, reg_val zero iff :none, d sum(y)/d(x, ps) finite and all non-zero, d reg_val/d x ===
    nothing, d reg_val/d ps finite with some non-zero.  The layer here is the ADAPTIVE one (the default)."""
    P = gpu_pkg
    node = P.NeuralDSDE(P.Chain(P.Dense(2, 4, "gelu"), P.Dense(4, 2)), P.Dense(2, 2), regularize=regularize, tspan=(0.0, 1.0),
                        nfine=512, abstol=1e-2, reltol=1e-2)
    assert node.adaptive
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((1, 2)).astype(f32)).cuda()
    pd, pg = _params(2, 4, 1)
    ps = dict(drift=pd, diffusion=pg)
    st = node.initialstates(np.random.default_rng(0))
    sol, st_ = node(x, ps, st)
    y = P.diffeqsol_to_array(sol)
    assert y.dtype == torch.float32 and y.shape == x.shape and torch.isfinite(y).all()          # :354
    assert (st_["reg_val"] == 0) == (regularize == "none")                                      # :355, :382, :415
    assert st_["nfe_drift"] > 0 and st_["nfe_diffusion"] > 0
    dx, dps, info = node.pullback(x, ps, st, torch.ones_like(x), w_reg=0.0)                     # gradient of sum(y)
    assert torch.isfinite(dx).all() and (dx != 0).all()                                         # :359-360
    for k in ("drift", "diffusion"):
        assert torch.isfinite(dps[k]).all() and (dps[k] != 0).all(), k                          # :361-362
    assert torch.equal(info["sol"].u[-1], y)
    if regularize != "none":                                                                    # :391-397: reg_val alone
        dxr, dpsr, infor = node.pullback(x, ps, st, torch.zeros_like(x), w_reg=1.0)
        assert infor["dx_reg"] is None and not (dxr != 0).any()                                 # gs_x === nothing
        for k in ("drift", "diffusion"):
            assert torch.isfinite(dpsr[k]).all() and (dpsr[k] != 0).any(), k
    # test mode: the vanilla fallback (:84, :107)
    sol_t, st_t = node(x, ps, dict(st, training=False))
    assert st_t["reg_val"] == 0


def test_layer_draws_are_reproducible_and_fixed_grid_mode_remains(gpu_pkg):
    P = gpu_pkg
    D, H, B = 32, 64, 8
    pd, pg = _params(D, H, 3)
    ps = dict(drift=pd, diffusion=pg)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((B, D)).astype(f32)).cuda()
    node = P.NeuralDSDE(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D)), P.Dense(D, D), abstol=0.14, reltol=0.14, nfine=64)
    st = node.initialstates(np.random.default_rng(0))
    a, sa = node(x, ps, st)
    b, sb = node(x, ps, st)
    assert torch.equal(a.u[-1], b.u[-1]) and sa["reg_val"] == sb["reg_val"] and sa["nfe_drift"] == sb["nfe_drift"]
    fixed = P.NeuralDSDE(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D)), P.Dense(D, D), abstol=0.14, reltol=0.14, adaptive=False, nsteps=8)
    c, sc = fixed(x, ps, fixed.initialstates(np.random.default_rng(0)))
    assert sc["nfe_drift"] == 3 * 8 + 3 and torch.isfinite(c.u[-1]).all()


import os as _os


@pytest.mark.parametrize("seed", list(range(int(_os.environ.get("LRNDE_SOAK_SEEDS", "8")))))
def test_adaptive_layer_soak_bit_exact(oracle, gpu_pkg, seed):
    """random shape (inside and outside the one-launch kernel's range), batch, grid, tolerance, mode, saveat list, save_start, t1
    draw, explicit or automatic initial dt: forward == the oracle loop bit for bit; LRNDE_SOAK_SEEDS=N runs N seeds"""
    rng = np.random.default_rng(90_000 + seed)
    D = int(rng.choice([1, 2, 7, 16, 32, 33, 48, 64, 70]))
    H = int(rng.choice([3, 16, 40, 64, 100, 113, 128, 130]))
    B = int(rng.choice([1, 3, 16, 17, 64, 130]))
    nfine = int(rng.choice([16, 64, 200]))
    tol = float(rng.choice([0.14, 0.05, 0.5]))
    mode = str(rng.choice(["unbiased", "biased", "none"]))
    act = str(rng.choice(["tanh", "gelu"]))
    nsv = int(rng.integers(0, 4))
    saveat = tuple(sorted(float(f32(v)) for v in rng.random(nsv))) if nsv else ()
    if nsv and rng.random() < 0.5:
        saveat = saveat + (1.0,)
    save_start = int(rng.choice([-1, 0, 1]))
    dt0 = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=seed, act=act, scale=float(rng.choice([1.0, 2.5])))
    kw = dict(mode=mode, t1_or_rand=float(f32(rng.random())), saveat=saveat, save_start=save_start, dt0=dt0)
    what = f"seed={seed} D={D} H={H} B={B} nfine={nfine} tol={tol} {act} {kw}"
    try:
        ref = oracle.sde_node_forward(drift, diff, x, W, 0.0, 1.0, tol, tol, z_local=z, **kw)
    except AssertionError as e:    # the oracle loop's own DtLessThanMin / ":biased needs two saved times": the library must refuse too
        with pytest.raises(gpu_pkg.LrndeError):
            h.node_forward_record(torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), 0.0, 1.0, tol, tol, z_local=torch.from_numpy(z).cuda(), **kw)
        return
    got = h.node_forward_record(torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), 0.0, 1.0, tol, tol, z_local=torch.from_numpy(z).cuda(), **kw)
    _check_forward(got, ref, what)


@pytest.mark.parametrize("D,H,B,tol,nfine,mode,saveat", [(32, 64, 512, 0.14, 128, "unbiased", ()), (20, 48, 33, 0.05, 64, "biased", ()),
                                                         (64, 128, 17, 0.1, 64, "unbiased", (0.3, 0.7, 1.0)), (2, 4, 1, 0.05, 64, "none", (0.0, 0.5, 1.0)),
                                                         (33, 100, 9, 0.1, 32, "unbiased", ())])
def test_one_launch_reverse_sweep_agrees_with_the_generic_path(oracle, gpu_pkg, D, H, B, tol, nfine, mode, saveat):
    """lrnde_sde_bwd_fused.hpp: the whole reverse sweep in one launch (a workgroup carries four samples through every recorded
    step) against the launch-per-piece path it replaces (LRNDE_NO_SDE_BWD_FUSED=1).  Different summation orders (plain fma chains
    and per-workgroup partials here, MFMA chains and the four-chain batch sum there): agreement to fp32 rounding of the sums."""
    h, drift, diff, pd, pg, x, W, z = _setup(gpu_pkg, oracle, D, H, B, nfine, seed=31, scale=1.5)
    xd, Wd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(z).cuda()
    kw = dict(mode=mode, t1_or_rand=0.37, saveat=saveat, save_start=-1)
    outs = []
    for flag in (0, 1):
        gpu_pkg.set_option("LRNDE_NO_SDE_BWD_FUSED", flag)
        try:
            got = h.node_forward_record(xd, Wd, 0.0, 1.0, tol, tol, z_local=zd, **kw)
            ns = got["u"].shape[0]
            du = np.random.default_rng(5).standard_normal((ns, B, D)).astype(f32)
            bw = h.node_backward_recorded(torch.from_numpy(du).cuda(), w_reg=2.0 if mode != "none" else 0.0)
        finally:
            gpu_pkg.set_option("LRNDE_NO_SDE_BWD_FUSED", 0)
        outs.append({k: bw[k].cpu().numpy() for k in ("dx", "dp_drift", "dp_diff")})
    for k in ("dx", "dp_drift", "dp_diff"):
        e = _rel(outs[0][k], outs[1][k])
        print(f"D={D} H={H} B={B} {mode}: fused vs generic {k} {e:.2e}")
        assert e < 2e-6, (k, e)


def test_diffusion_without_a_bias_through_both_reverse_sweeps(gpu_pkg):
    """Dense(D => D, use_bias=false) as the diffusion: the parameter vector is vec(Wg) alone; the one-launch sweep and the generic
    one agree and leave D*D cotangents"""
    from localregneuralde_jl_amd.layers import _mlp_desc
    D, H, B, nfine = 12, 20, 9, 32
    rng = np.random.default_rng(3)
    pd = (rng.standard_normal(D * H + H + H * D + D) * 0.3).astype(f32)
    pg = (rng.standard_normal(D * D) * 0.1).astype(f32)
    h = gpu_pkg.SdeHandle(_mlp_desc(gpu_pkg.Chain(gpu_pkg.Dense(D, H, "tanh"), gpu_pkg.Dense(H, D))), diffusion_bias=False)
    h.set_params(pd, pg)
    x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
    W = torch.from_numpy(_path(rng, nfine, B, D)).cuda()
    z = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
    outs = []
    for flag in (0, 1):
        gpu_pkg.set_option("LRNDE_NO_SDE_BWD_FUSED", flag)
        try:
            fw = h.node_forward_record(x, W, 0.0, 1.0, 0.05, 0.05, z_local=z, mode="unbiased", t1_or_rand=0.4, saveat=(), save_start=-1)
            du = torch.from_numpy(np.random.default_rng(5).standard_normal((fw["u"].shape[0], B, D)).astype(f32)).cuda()
            bw = h.node_backward_recorded(du, w_reg=1.5)
        finally:
            gpu_pkg.set_option("LRNDE_NO_SDE_BWD_FUSED", 0)
        outs.append({k: bw[k].cpu().numpy() for k in ("dx", "dp_drift", "dp_diff")})
    assert outs[0]["dp_diff"].shape == (D * D,)
    for k in ("dx", "dp_drift", "dp_diff"):
        assert np.isfinite(outs[0][k]).all() and _rel(outs[0][k], outs[1][k]) < 2e-6, (k, _rel(outs[0][k], outs[1][k]))
