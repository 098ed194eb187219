"""Pins for the oracle's CIFAR10 stem and head (experiments/src/construct.jl:224-227, src/layers/common.jl:80-92) against
torch float64 autograd."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def _gelu(z):
    return 0.5 * z * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))


@pytest.mark.parametrize("train", [True, False])
def test_stem_matches_torch(train):
    rng = np.random.default_rng(0)
    B, H, W = 3, 6, 8
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    ps = (rng.standard_normal(156) * 0.3).astype(np.float32)
    ps[140:148] = rng.uniform(0.5, 1.5, 8)
    st = None if train else np.concatenate([rng.normal(0, 0.2, 8), rng.uniform(0.5, 2, 8)]).astype(np.float32)
    g = rng.standard_normal((B, 8, H, W)).astype(np.float32)
    u0 = O.cifar_stem_forward(x, ps, bn_train=train, bn_state=st)
    dps = O.cifar_stem_backward(x, ps, g, bn_train=train, bn_state=st)
    pt = torch.tensor(ps.astype(np.float64), requires_grad=True)
    xt = torch.tensor(x.astype(np.float64))
    w = torch.flip(pt[:135].reshape(5, 3, 3, 3), dims=(2, 3))  # (kx,ky,ci,co) column-major -> (co,ci,ky,kx), NNlib.conv flips
    a0 = torch.cat([xt, torch.nn.functional.conv2d(xt, w, bias=pt[135:140], padding=1)], dim=1)
    if train:
        mu = a0.mean(dim=(0, 2, 3), keepdim=True); var = a0.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mu = torch.tensor(st[:8].astype(np.float64)).reshape(1, 8, 1, 1); var = torch.tensor(st[8:].astype(np.float64)).reshape(1, 8, 1, 1)
    out = (a0 - mu) / torch.sqrt(var + 1e-5) * pt[140:148].reshape(1, 8, 1, 1) + pt[148:156].reshape(1, 8, 1, 1)
    np.testing.assert_allclose(u0, out.detach().numpy(), rtol=0, atol=2e-5 * np.abs(out.detach().numpy()).max())
    (out * torch.tensor(g.astype(np.float64))).sum().backward()
    ref = pt.grad.numpy()
    assert np.abs(dps - ref).max() <= 2e-5 * np.abs(ref).max()


def test_stem_running_statistics_match_torch_batch_norm():
    """Lux's training-mode BatchNorm returns advanced running statistics in its state (UPSTREAM-RECALL: momentum 0.1,
    n/(n-1) variance correction) — the rule torch.nn.functional.batch_norm applies to its running buffers"""
    rng = np.random.default_rng(3)
    B, H, W = 4, 6, 8
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    ps = (rng.standard_normal(156) * 0.3).astype(np.float32)
    st0 = np.concatenate([rng.normal(0, 0.2, 8), rng.uniform(0.5, 2, 8)]).astype(np.float32)
    for st in (None, st0):
        u0, st1 = O.cifar_stem_forward(x, ps, bn_train=True, bn_state=st, return_state=True)
        xt = torch.tensor(x.astype(np.float64)); pt = torch.tensor(ps.astype(np.float64))
        w = torch.flip(pt[:135].reshape(5, 3, 3, 3), dims=(2, 3))
        a0 = torch.cat([xt, torch.nn.functional.conv2d(xt, w, bias=pt[135:140], padding=1)], dim=1)
        rm = torch.zeros(8, dtype=torch.float64) if st is None else torch.tensor(st[:8].astype(np.float64))
        rv = torch.ones(8, dtype=torch.float64) if st is None else torch.tensor(st[8:].astype(np.float64))
        out = torch.nn.functional.batch_norm(a0, rm, rv, pt[140:148], pt[148:156], training=True, momentum=0.1, eps=1e-5)
        np.testing.assert_allclose(u0, out.numpy(), rtol=0, atol=2e-5 * np.abs(out.numpy()).max())
        np.testing.assert_allclose(st1[:8], rm.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st1[8:], rv.numpy(), rtol=1e-5, atol=1e-6)
    # test mode: the state passes through unchanged
    _, st2 = O.cifar_stem_forward(x, ps, bn_train=False, bn_state=st0, return_state=True)
    np.testing.assert_array_equal(st2, st0)


def test_head_matches_torch():
    rng = np.random.default_rng(1)
    B, H, W, K = 4, 6, 8, 10
    u = rng.standard_normal((B, 8, H, W)).astype(np.float32)
    n = O.lib().lro_cifar_head_param_count(H, W, K)
    assert n == 72 + 1 + K * H * W + K
    ph = (rng.standard_normal(n) * 0.2).astype(np.float32)
    lab = rng.integers(0, K, B)
    loss, logits, du, dph = O.cifar_head_ce(u, ph, K, lab)
    pt = torch.tensor(ph.astype(np.float64), requires_grad=True)
    ut = torch.tensor(u.astype(np.float64), requires_grad=True)
    w = torch.flip(pt[:72].reshape(1, 8, 3, 3), dims=(2, 3))
    v = _gelu(torch.nn.functional.conv2d(ut, w, bias=pt[72:73], padding=1)).reshape(B, H * W)  # Julia flatten of (W,H,1,B): w fastest
    Wd = pt[73:73 + K * H * W].reshape(H * W, K).t()
    lg = v @ Wd.t() + pt[73 + K * H * W:]
    ce = torch.nn.functional.cross_entropy(lg, torch.tensor(lab))
    ce.backward()
    assert abs(float(loss) - ce.item()) < 2e-6 * max(1.0, abs(ce.item()))
    np.testing.assert_allclose(logits, lg.detach().numpy(), rtol=0, atol=2e-5 * np.abs(lg.detach().numpy()).max())
    assert np.abs(du - ut.grad.numpy()).max() <= 2e-5 * np.abs(ut.grad.numpy()).max()
    assert np.abs(dph - pt.grad.numpy()).max() <= 2e-5 * np.abs(pt.grad.numpy()).max()
