"""The layer forward runs its local step — and, recording, the regulariser's reverse sweep — on the handle's companion
stream while the main solve is still integrating [t1, t2].  Same kernels on the same inputs: every
output must be the SAME BITS as with everything in order on one stream (lrnde_set_overlap(0), include/lrnde_hooks.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(P, B, seed):
    import torch
    D, H, K = 784, 100, 10
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    rng = np.random.default_rng(seed)
    ps = torch.from_numpy(P.glorot_params(model, seed=seed)).cuda()
    pc = torch.from_numpy((rng.standard_normal(K * (D + 1)) * 0.05).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
    return model, ps, pc, x, lab


@pytest.mark.parametrize("t1", [0.03, 0.5, 0.97, 1.0])   # early, middle, inside the last steps, t1 == t2
@pytest.mark.parametrize("reg_type", ["error_estimate", "stiffness_estimate"])
def test_forward_same_bits_with_and_without_the_companion_stream(gpu_pkg, t1, reg_type):
    import torch
    P = gpu_pkg
    model, ps, _, x, _ = _setup(P, 96, 3)
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    h = Handle(_mlp_desc(model)); h.set_params(ps)
    out = []
    for on in (True, False, True):
        h.set_overlap(on)
        out.append(h.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, mode="unbiased", reg_type=reg_type, t1_or_rand=t1, maxiters=2000))
    for o in out[1:]:
        assert torch.equal(o["u_end"], out[0]["u_end"])
        assert o["reg_val"] == out[0]["reg_val"] and o["nfe"] == out[0]["nfe"] and o["stats"] == out[0]["stats"]


@pytest.mark.parametrize("saveat", [None, [0.25, 0.5, 1.0]])
def test_training_step_same_bits_with_and_without_the_companion_stream(gpu_pkg, saveat):
    import torch
    P = gpu_pkg
    model, ps, pc, x, lab = _setup(P, 64, 5)
    res = []
    for on in (True, False):
        kw = {} if saveat is None else {"saveat": saveat}
        node = P.NeuralODE(model, regularize="unbiased", abstol=1e-5, reltol=1e-5, save_start=False, maxiters=2000, **kw)
        st = node.initialstates(np.random.default_rng(0))
        node._bind(ps, x).set_overlap(on)
        if saveat is None:
            loss, _, _, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
            res.append((loss, grads))
        else:   # the layer's saveat kwarg: one cotangent per state of the series (src/utils.jl:25-46)
            du = torch.stack([torch.full_like(x, 0.01 * (i + 1)) for i in range(len(saveat))])
            dx, dp, info = node.pullback(x, ps, st, du, w_reg=2.5)
            res.append((float(info["reg_val"]), {"dx": dx, "dp": dp}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("where", ["early", "late"])
def test_training_step_with_an_early_and_a_late_t1(gpu_pkg, where):
    """early t1: the sweep is enqueued beside the forward solve; late t1 (sol(t1) appears in the last reports): the forward
    leaves it to the backward pass, which enqueues it on the companion's stream after its first adjoint attempt.  Either
    way the gradients are the bits of the one-stream order."""
    import torch
    P = gpu_pkg
    model, ps, pc, x, lab = _setup(P, 64, 7)
    def r01_of(sd):  # the draw run_training_step makes from initialstates(default_rng(sd))
        g = np.random.default_rng(sd); g.standard_normal()
        return float(g.random(dtype=np.float32))
    seed = next(sd for sd in range(2000) if (r01_of(sd) > 0.95 if where == "late" else r01_of(sd) < 0.08))
    res = []
    for on in (True, False, True):
        node = P.NeuralODE(model, regularize="unbiased", abstol=1e-5, reltol=1e-5, save_start=False, maxiters=2000)
        st = node.initialstates(np.random.default_rng(seed))
        node._bind(ps, x).set_overlap(on)
        loss, st2, _, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
        res.append((loss, grads))
    for r in res[1:]:
        assert r[0] == res[0][0]
        for k in res[0][1]:
            assert torch.equal(r[1][k], res[0][1][k]), k


@pytest.mark.parametrize("save_start", [True, False])
@pytest.mark.parametrize("t1", [0.2, 0.8])
def test_recorded_forward_with_save_start_uses_the_right_slot(oracle, gpu_pkg, save_start, t1):
    """With save_start the solution's first slot is u(t0) and sol(t1) sits one slot later: the companion stream's local step (and
    the regulariser sweep it feeds) must start from sol(t1), not from the start value.  (A wrong slot was found by the
    randomised soak: forward values were right — the serial path recomputed them — but the recorded regulariser gradient came
    from a local step at u(t0).)  Pullback: both stream orders bit for bit, and the oracle's within tolerance."""
    import torch
    from test_gpu_backward import _mk, _rel
    P = gpu_pkg
    fld, h, p, x = _mk(oracle, P, 784, 100, 16, "tanh", True, scale=1.5)
    g = (np.random.default_rng(4).standard_normal(x.shape) * 1e-2).astype(np.float32)
    xd, gd = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    ref = oracle.node_backward(fld, x, 0.0, 1.0, 1e-5, 1e-5, g, mode="unbiased", t1_or_rand=t1, w_reg=2.5, save_start=save_start)
    out = []
    for on in (True, False):
        h.set_overlap(on)
        out.append(h.node_backward(xd, 0.0, 1.0, 1e-5, 1e-5, gd, mode="unbiased", t1_or_rand=t1, w_reg=2.5, maxiters=10000,
                                   save_start=save_start))
    assert torch.equal(out[0]["dx"], out[1]["dx"]) and torch.equal(out[0]["dp"], out[1]["dp"])
    assert _rel(out[0]["dx"].cpu().numpy(), ref["dx"]) < 2e-5 and _rel(out[0]["dp"].cpu().numpy(), ref["dp"]) < 3e-4


@pytest.mark.parametrize("t1", [0.04, 0.5, 0.96])
def test_solve_loop_without_reports_gives_the_same_bits(gpu_pkg, t1):
    """`lrnde_set_reports(ctx, 0)`: the solve loop's fall-back (polled copies of the control block instead of the per-launch
    reports in pinned host memory) against the report-driven loop — plain solve with a trace, layer forward, recorded
    forward + head + backward"""
    import numpy as np
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    P = gpu_pkg
    D, H, B, K = 784, 100, 40, 10
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    p = torch.from_numpy(P.glorot_params(model, seed=1) * np.float32(1.5))
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
    pc = torch.from_numpy((rng.random(K * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).cuda()
    lab = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
    ha, hb = Handle(_mlp_desc(model)), Handle(_mlp_desc(model))
    ha.set_params(p); hb.set_params(p)
    hb.set_reports(False)
    sa = ha.solve(x, 0.0, 1.0, 1e-6, 1e-6, saveat=[0.3, t1, 1.0] if t1 > 0.3 else [t1, 0.3, 1.0], maxiters=10000, trace=True)
    sb = hb.solve(x, 0.0, 1.0, 1e-6, 1e-6, saveat=[0.3, t1, 1.0] if t1 > 0.3 else [t1, 0.3, 1.0], maxiters=10000, trace=True)
    assert sa["stats"] == sb["stats"] and torch.equal(sa["u"], sb["u"]) and np.array_equal(sa["t"], sb["t"])
    assert np.array_equal(sa["trace"], sb["trace"])
    kw = dict(mode="unbiased", reg_type="stiffness_estimate", t1_or_rand=t1, maxiters=10000, save_start=True)
    fa = ha.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, **kw); fb = hb.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, **kw)
    assert torch.equal(fa["u_end"], fb["u_end"]) and fa["reg_val"] == fb["reg_val"] and fa["nfe"] == fb["nfe"] and fa["stats"] == fb["stats"]
    ra, qa = ha.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, lab, **kw)
    rb, qb = hb.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, lab, **kw)
    assert torch.equal(ra["u_end"], rb["u_end"]) and ra["reg_val"] == rb["reg_val"] and qa["loss"] == qb["loss"] and torch.equal(qa["du"], qb["du"])
    ba = ha.node_backward_recorded(qa["du"], w_reg=2.0); bb = hb.node_backward_recorded(qb["du"], w_reg=2.0)
    assert torch.equal(ba["dx"], bb["dx"]) and torch.equal(ba["dp"], bb["dp"])
    with pytest.raises(P.LrndeError):     # failures come back through the fall-back loop as well
        hb.solve(x, 0.0, 1.0, 1e-7, 1e-7, saveat=[1.0], maxiters=3)
