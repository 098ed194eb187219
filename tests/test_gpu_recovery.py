"""A failed call leaves the handle usable.  The layer forward runs its local step (and, recording, the regulariser's reverse
sweep) on the handle's companion stream while the main solve is still integrating: a solve that stops with
MAXITERS or DtNaN does so with companion work in flight.  After every failure the same handle must give, bit for bit, what a
fresh handle gives — forward, record and backward — and a backward without a valid record must be refused, not run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(P, D=784, H=100, B=48, seed=0):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    p = torch.from_numpy(P.glorot_params(model, seed=seed) * np.float32(1.5))
    x = torch.from_numpy(np.random.default_rng(seed + 1).random((B, D), dtype=np.float32)).cuda()
    hs = []
    for _ in range(2):
        h = Handle(_mlp_desc(model)); h.set_params(p); hs.append(h)
    return hs[0], hs[1], x


def _same(a, b):
    import torch
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


@pytest.mark.parametrize("t1", [0.03, 0.43, 0.97])
@pytest.mark.parametrize("reg_type", ["error_estimate", "stiffness_estimate"])
def test_forward_and_backward_after_failures(gpu_pkg, t1, reg_type):
    import torch
    P = gpu_pkg
    h, fresh, x = _pair(P)
    kw = dict(mode="unbiased", reg_type=reg_type, t1_or_rand=t1)
    du = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(x.shape)).astype(np.float32)).cuda()
    want_f = fresh.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw)
    want_r = fresh.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw)
    want_b = fresh.node_backward_recorded(du, w_reg=2.5)

    # 1. MAXITERS in the plain forward (the local step at t1 = 0.03 is already running beside it)
    with pytest.raises(P.LrndeError) as e:
        h.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=4, **kw)
    assert e.value.code == 1, e.value
    _same(h.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw), want_f)

    # 2. MAXITERS in the recording forward: no record is left behind, the backward is refused
    with pytest.raises(P.LrndeError) as e:
        h.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=4, **kw)
    assert e.value.code == 1, e.value
    with pytest.raises(P.LrndeError):
        h.node_backward_recorded(du, w_reg=2.5)
    _same(h.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw), want_r)
    _same(h.node_backward_recorded(du, w_reg=2.5), want_b)
    with pytest.raises(P.LrndeError):                      # the record is consumed by its backward
        h.node_backward_recorded(du, w_reg=2.5)

    # 3. a NaN in the input: DtNaN from the step controller, then business as usual
    xn = x.clone(); xn[3, 5] = float("nan")
    with pytest.raises(P.LrndeError) as e:
        h.node_forward_record(xn, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw)
    assert e.value.code == 3, e.value
    _same(h.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw), want_r)
    _same(h.node_backward_recorded(du, w_reg=2.5), want_b)

    # 4. a NaN cotangent: the adjoint stops with an error status, the next forward/backward pair is clean
    h.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw)
    dn = du.clone(); dn[0, 0] = float("nan")
    try:
        got = h.node_backward_recorded(dn, w_reg=2.5)
        assert not torch.isfinite(got["dx"]).all()        # (or NaN gradients handed back: the caller sees them)
    except P.LrndeError:
        pass
    _same(h.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw), want_r)
    _same(h.node_backward_recorded(du, w_reg=2.5), want_b)
    _same(h.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, maxiters=10000, **kw), want_f)


@pytest.mark.gpu
def test_record_generation_tells_a_stale_record_from_the_callers_own(gpu_pkg):
    """a binding's pullback closure must be able to see that ANOTHER forward of the same layer replaced the record
    it was made for (the C side can only see that some record is valid).  lrnde_record_generation counts the recorded
    forwards; 0 = no usable record (none yet, or consumed by a backward)."""
    import numpy as np
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    P = gpu_pkg
    D, H, B = 32, 16, 8
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(P.glorot_params(model, seed=0)))
    x = torch.from_numpy(np.random.default_rng(0).random((B, D), dtype=np.float32)).cuda()
    assert h.record_generation() == 0
    h.node_forward_record(x, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.3)
    g1 = h.record_generation()
    assert g1 == 1
    h.node_forward_record(x * 2, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.3)   # "an evaluation pass in between"
    assert h.record_generation() == 2 != g1          # the first closure sees that its record is gone
    h.node_backward_recorded(torch.ones_like(x), w_reg=1.0)
    assert h.record_generation() == 0                # consumed
    h.node_forward(x, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.3)               # an unrecorded forward makes none
    assert h.record_generation() == 0
