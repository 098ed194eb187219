"""Pins for the oracle's conv vector field (experiments/src/construct.jl:213-218): an independent
torch-CPU float64 restatement (conv2d with the kernel flipped = NNlib.conv, the t plane concatenated
as the last channel before every conv, train/test-mode batch norm, tanh-form gelu)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def torch_conv_field(u, t, params, W, H, C, Hc, act, bn_train, bn_state=None, eps=1e-5):
    B = u.shape[0]
    x = torch.from_numpy(u.astype(np.float64)).reshape(B, C, H, W)  # Julia WHCN == torch NCHW memory order
    p = torch.from_numpy(params.astype(np.float64))
    off = 0

    def take(n):
        nonlocal off
        v = p[off:off + n]
        off += n
        return v

    def weight(cin, cout):
        # Julia (kx,ky,ci,co) column-major -> torch (co,ci,ky,kx); NNlib.conv flips the kernel
        w = take(9 * cin * cout).reshape(cout, cin, 3, 3)
        return torch.flip(w, dims=(2, 3))

    def tcat(z):
        return torch.cat([z, torch.full((B, 1, H, W), float(t), dtype=torch.float64)], dim=1)

    def fact(z):
        if act == "gelu":
            return 0.5 * z * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))
        if act == "tanh":
            return torch.tanh(z)
        return z

    def bn(z, k):
        g, b = take(Hc), take(Hc)
        if bn_train:
            mu = z.mean(dim=(0, 2, 3), keepdim=True)
            var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        else:
            st = np.zeros(4 * Hc) if bn_state is None else bn_state.astype(np.float64)
            if bn_state is None:
                st[Hc:2 * Hc] = 1.0
                st[3 * Hc:] = 1.0
            mu = torch.from_numpy(st[2 * k * Hc:(2 * k + 1) * Hc]).reshape(1, Hc, 1, 1)
            var = torch.from_numpy(st[(2 * k + 1) * Hc:(2 * k + 2) * Hc]).reshape(1, Hc, 1, 1)
        return fact((z - mu) / torch.sqrt(var + eps) * g.reshape(1, Hc, 1, 1) + b.reshape(1, Hc, 1, 1))

    z = torch.nn.functional.conv2d(tcat(x), weight(C + 1, Hc), padding=1)
    z = bn(z, 0)
    z = torch.nn.functional.conv2d(tcat(z), weight(Hc + 1, Hc), padding=1)
    z = bn(z, 1)
    z = torch.nn.functional.conv2d(tcat(z), weight(Hc + 1, C), padding=1)
    assert off == p.numel()
    return z.reshape(B, -1).numpy()


def _case(W, H, C, Hc, B, seed, scale=1.0, bn_affine=True):
    rng = np.random.default_rng(seed)
    p = O.glorot_conv_params(C, Hc, seed=seed) * np.float32(scale)
    if bn_affine:  # non-trivial scale / bias so the test sees them
        n1 = 9 * (C + 1) * Hc
        p[n1:n1 + Hc] = rng.uniform(0.5, 1.5, Hc)
        p[n1 + Hc:n1 + 2 * Hc] = rng.uniform(-0.3, 0.3, Hc)
        n2 = n1 + 2 * Hc + 9 * (Hc + 1) * Hc
        p[n2:n2 + Hc] = rng.uniform(0.5, 1.5, Hc)
        p[n2 + Hc:n2 + 2 * Hc] = rng.uniform(-0.3, 0.3, Hc)
    u = rng.standard_normal((B, W * H * C)).astype(np.float32)
    return p.astype(np.float32), u


@pytest.mark.parametrize("W,H,C,Hc,B,train", [(8, 8, 8, 16, 3, True), (8, 6, 4, 8, 2, False), (12, 12, 8, 64, 2, True)])
@pytest.mark.parametrize("act", ["gelu", "tanh"])
def test_conv_field_matches_torch_float64(W, H, C, Hc, B, train, act):
    p, u = _case(W, H, C, Hc, B, seed=W + Hc)
    st = None
    if not train:
        rng = np.random.default_rng(5)
        st = np.concatenate([rng.normal(0, 0.2, Hc), rng.uniform(0.5, 2, Hc), rng.normal(0, 0.2, Hc),
                             rng.uniform(0.5, 2, Hc)]).astype(np.float32)
    fld = O.ConvField(W, H, C, Hc, p, act=act, bn_train=train, bn_state=st, nthreads=4)
    got = fld.rhs(u, 0.37)
    ref = torch_conv_field(u, 0.37, p, W, H, C, Hc, act, train, st)
    sc = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * sc  # fp32 chains vs float64


def test_conv_param_count_cifar_block():
    # 3*3*9*64 + 2*64 + 3*3*65*64 + 2*64 + 3*3*65*8
    assert O.lib().lro_conv_param_count(8, 64) == 5184 + 128 + 37440 + 128 + 4680


def test_t_plane_is_zero_padded_like_a_channel():
    """The t plane is concatenated before the conv (src/layers/common.jl:10-45), so at the image border
    fewer t taps contribute: with all weights zero except the t channel of conv3, du = t * (number of
    in-image taps) * w."""
    W = H = 5; C = 2; Hc = 4
    p = np.zeros(O.lib().lro_conv_param_count(C, Hc), np.float32)
    n3 = 9 * (C + 1) * Hc + 2 * Hc + 9 * (Hc + 1) * Hc + 2 * Hc
    w3 = p[n3:].reshape(C, Hc + 1, 3, 3)  # (co, ci, ky, kx) view of the column-major (kx,ky,ci,co)
    w3[:, Hc, :, :] = 1.0
    fld = O.ConvField(W, H, C, Hc, p, bn_train=False)
    du = fld.rhs(np.zeros((1, W * H * C), np.float32), 2.0).reshape(C, H, W)
    cnt = np.full((H, W), 9.0); cnt[0, :] = 6; cnt[-1, :] = 6; cnt[:, 0] = 6; cnt[:, -1] = 6
    cnt[0, 0] = cnt[0, -1] = cnt[-1, 0] = cnt[-1, -1] = 4
    assert np.array_equal(du[0], 2.0 * cnt) and np.array_equal(du[1], 2.0 * cnt)


def test_conv_field_drives_the_generic_solver():
    """the oracle's Tsit5 machinery is field-agnostic: a conv-field solve converges with tolerance"""
    W = H = 8; C = 8; Hc = 16; B = 2
    p, u = _case(W, H, C, Hc, B, seed=3, bn_affine=False)
    fld = O.ConvField(W, H, C, Hc, p, nthreads=4)
    a = O.solve(fld, u, 0.0, 1.0, 1e-3, 1e-3)
    b = O.solve(fld, u, 0.0, 1.0, 1e-6, 1e-6)
    assert a["stats"]["retcode"] == 0 and b["stats"]["retcode"] == 0
    assert b["stats"]["naccept"] > a["stats"]["naccept"]
    ua, ub = a["u"][-1], b["u"][-1]
    assert np.abs(ua - ub).max() < 5e-3 * max(1.0, np.abs(ub).max())


@pytest.mark.parametrize("W,H,C,Hc,B,train,act", [(8, 8, 8, 16, 3, True, "gelu"), (8, 6, 4, 8, 2, False, "gelu"),
                                                  (6, 6, 8, 64, 2, True, "tanh")])
def test_conv_vjp_matches_torch_autograd(W, H, C, Hc, B, train, act):
    """lro_conv_vjp = what Zygote.pullback(dudt, y, p, t) returns (dy, dp), batch statistics differentiated"""
    p, u = _case(W, H, C, Hc, B, seed=W + Hc + 1)
    st = None
    if not train:
        rng = np.random.default_rng(5)
        st = np.concatenate([rng.normal(0, 0.2, Hc), rng.uniform(0.5, 2, Hc), rng.normal(0, 0.2, Hc),
                             rng.uniform(0.5, 2, Hc)]).astype(np.float32)
    lam = np.random.default_rng(2).standard_normal(u.shape).astype(np.float32)
    fld = O.ConvField(W, H, C, Hc, p, act=act, bn_train=train, bn_state=st, nthreads=4)
    dy, gp = O.conv_vjp(fld, u, 0.37, lam)

    # float64 autograd through the same restatement
    ut = torch.tensor(u.astype(np.float64), requires_grad=True)
    pt = torch.tensor(p.astype(np.float64), requires_grad=True)

    def field(uu, pp):
        x = uu.reshape(B, C, H, W)
        off = [0]

        def take(n):
            v = pp[off[0]:off[0] + n]; off[0] += n
            return v

        def weight(cin, cout):
            return torch.flip(take(9 * cin * cout).reshape(cout, cin, 3, 3), dims=(2, 3))

        def tcat(z):
            return torch.cat([z, torch.full((B, 1, H, W), 0.37, dtype=torch.float64)], dim=1)

        def fact(z):
            if act == "gelu":
                return 0.5 * z * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))
            return torch.tanh(z)

        def bn(z, k):
            g, b = take(Hc), take(Hc)
            if train:
                mu = z.mean(dim=(0, 2, 3), keepdim=True); var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            else:
                s64 = torch.tensor(st.astype(np.float64))
                mu = s64[2 * k * Hc:(2 * k + 1) * Hc].reshape(1, Hc, 1, 1); var = s64[(2 * k + 1) * Hc:(2 * k + 2) * Hc].reshape(1, Hc, 1, 1)
            return fact((z - mu) / torch.sqrt(var + 1e-5) * g.reshape(1, Hc, 1, 1) + b.reshape(1, Hc, 1, 1))

        z = bn(torch.nn.functional.conv2d(tcat(x), weight(C + 1, Hc), padding=1), 0)
        z = bn(torch.nn.functional.conv2d(tcat(z), weight(Hc + 1, Hc), padding=1), 1)
        return torch.nn.functional.conv2d(tcat(z), weight(Hc + 1, C), padding=1).reshape(B, -1)

    out = field(ut, pt)
    (out * torch.tensor(lam.astype(np.float64))).sum().backward()
    dy_ref, gp_ref = ut.grad.numpy(), pt.grad.numpy()
    assert np.abs(dy - dy_ref).max() <= 3e-5 * np.abs(dy_ref).max()
    assert np.abs(gp - gp_ref).max() <= 3e-5 * np.abs(gp_ref).max()


def test_conv_node_backward_matches_finite_differences():
    """the field-agnostic backward drivers with the conv field: directional derivative of
    L = <w, sol.u[end]> + w_reg * reg_val along a random parameter direction vs central differences (fp32 solve,
    so only a loose agreement is asked; the drivers themselves are pinned by the MLP autograd tests)"""
    W = H = 6; C = 4; Hc = 8; B = 2
    p, u = _case(W, H, C, Hc, B, seed=4)
    rng = np.random.default_rng(0)
    wv = rng.standard_normal(u.shape).astype(np.float32)
    d = rng.standard_normal(p.shape).astype(np.float32)
    tol, w_reg, t1 = 1e-6, 0.0, 0.4

    def loss(pp):
        fld = O.ConvField(W, H, C, Hc, pp.astype(np.float32), nthreads=2)
        r = O.node_forward(fld, u, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=t1, maxiters=5000)
        return float((r["u_end"].astype(np.float64) * wv).sum()) + w_reg * float(r["reg_val"])

    fld = O.ConvField(W, H, C, Hc, p, nthreads=2)
    b = O.node_backward(fld, u, 0.0, 1.0, tol, tol, wv, mode="unbiased", t1_or_rand=t1, w_reg=w_reg, maxiters=5000)
    assert b["retcode"] == 0
    eps = 2e-3
    fd = (loss(p + eps * d) - loss(p - eps * d)) / (2 * eps)
    an = float((b["dp"].astype(np.float64) * d).sum())
    assert abs(fd - an) <= 2e-2 * max(abs(fd), abs(an)) + 1e-4


def test_conv_golden_fixture():
    """regression pin generated by tests/golden/make_golden.py from this oracle (conv_block_8x8_b2.npz)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "conv_block_8x8_b2.npz"))
    fld = O.ConvField(8, 8, 8, 64, g["params"], act="gelu", bn_train=True, nthreads=3)  # thread count must not matter
    assert np.array_equal(fld.rhs(g["x"], float(g["t"])), g["du"])
    dy, gp = O.conv_vjp(fld, g["x"], float(g["t"]), g["lam"])
    assert np.array_equal(dy, g["vjp_dy"]) and np.array_equal(gp, g["vjp_gp"])
    nd = O.node_forward(fld, g["x"], 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.41)
    assert nd["nfe"] == int(g["node_nfe"]) and nd["stats"]["naccept"] == int(g["node_naccept"])
    assert np.array_equal(nd["u_end"], g["node_u_end"]) and nd["reg_val"] == g["node_reg_val"]


def test_running_statistics_follow_torch_batchnorm():
    """training-mode BatchNorm advances its running statistics on every f-eval (what the reference's dudt closure does
    to its captured st_): momentum 0.1, unbiased variance — torch.nn.functional.batch_norm has the same rule"""
    W = H = 6; C = 8; Hc = 16; B = 3
    p, u = _case(W, H, C, Hc, B, seed=6)
    fld = O.ConvField(W, H, C, Hc, p, act="gelu", bn_train=True, nthreads=2)
    fld.rhs(u, 0.2); fld.rhs(u * np.float32(0.5), 0.7)
    # float64 restatement with torch's batch_norm keeping the running statistics
    rm = [torch.zeros(Hc, dtype=torch.float64) for _ in range(2)]
    rv = [torch.ones(Hc, dtype=torch.float64) for _ in range(2)]
    pt = torch.from_numpy(p.astype(np.float64))

    def run(uu, t):
        off = [0]

        def take(n):
            v = pt[off[0]:off[0] + n]; off[0] += n
            return v

        x = torch.from_numpy(uu.astype(np.float64)).reshape(B, C, H, W)
        tc = lambda z: torch.cat([z, torch.full((B, 1, H, W), t, dtype=torch.float64)], dim=1)
        wgt = lambda ci, co: torch.flip(take(9 * ci * co).reshape(co, ci, 3, 3), dims=(2, 3))
        gelu = lambda z: 0.5 * z * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))
        z = torch.nn.functional.conv2d(tc(x), wgt(C + 1, Hc), padding=1)
        g, b = take(Hc), take(Hc)
        z = gelu(torch.nn.functional.batch_norm(z, rm[0], rv[0], g, b, training=True, momentum=0.1, eps=1e-5))
        z = torch.nn.functional.conv2d(tc(z), wgt(Hc + 1, Hc), padding=1)
        g, b = take(Hc), take(Hc)
        torch.nn.functional.batch_norm(z, rm[1], rv[1], g, b, training=True, momentum=0.1, eps=1e-5)

    run(u, 0.2); run(u * np.float32(0.5), 0.7)
    ref = np.concatenate([rm[0].numpy(), rv[0].numpy(), rm[1].numpy(), rv[1].numpy()])
    np.testing.assert_allclose(fld.bn_run, ref, rtol=2e-5, atol=1e-6)
    # the VJP's recomputation leaves them alone
    before = fld.bn_run.copy()
    O.conv_vjp(fld, u, 0.3, u)
    assert np.array_equal(fld.bn_run, before)
