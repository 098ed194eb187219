"""End to end: the reference's training loop on the device (experiments/src/utils.jl:104-123 run_training_step + an Optimisers
update per step, experiments/src/construct.jl:104-126) — forward with record, classifier + logitcrossentropy, continuous
adjoint, regulariser gradient, Adam — on a small synthetic classification task.  The loss has to fall, every step has to
succeed (t1 is redrawn every step), and the run with the companion stream must end with the same parameters, bit for bit,
as the run with everything in order on one stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_a_short_training_run_learns_and_is_independent_of_the_stream_order(gpu_pkg):
    import torch
    P = gpu_pkg
    D, H, K, B, NSTEP = 784, 100, 10, 64, 80
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    centers = np.random.default_rng(0).random((K, D), dtype=np.float32)

    def batch(i):
        g = np.random.default_rng(1000 + i)
        lab = g.integers(0, K, B).astype(np.int32)
        x = (centers[lab] + 0.15 * g.standard_normal((B, D)).astype(np.float32)).clip(0, 1).astype(np.float32)
        return torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()

    def run(overlap):
        node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=1e-4, reltol=1e-4,
                           save_start=False, maxiters=10000)
        ps = torch.from_numpy(P.glorot_params(model, seed=0)).cuda()
        pc = torch.from_numpy((np.random.default_rng(2).random(K * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).cuda()
        opt = P.Optimiser("adam", learning_rate=1e-3)
        st = node.initialstates(np.random.default_rng(3))
        node._bind(ps, None).set_overlap(overlap)
        losses = []
        for i in range(NSTEP):
            x, lab = batch(i)
            loss, st, stats, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
            assert np.isfinite(float(loss)) and stats["nfe"] > 0
            opt.update([ps, pc], [grads["neural_ode"], grads["classifier"]])
            losses.append(float(loss))
        return ps.clone(), pc.clone(), losses

    pa, ca, la = run(True)
    pb, cb, lb = run(False)
    print(f"loss {la[0]:.4f} -> {np.mean(la[-5:]):.4f} in {NSTEP} steps")
    assert np.mean(la[-5:]) < 0.2 * la[0]
    assert la == lb and torch.equal(pa, pb) and torch.equal(ca, cb)


@pytest.mark.parametrize("mode,save_start,t1", [("unbiased", False, 0.43), ("unbiased", True, 0.02), ("unbiased", True, 0.97),
                                                ("biased", False, 0.6), ("none", True, 0.5)])
def test_fused_forward_and_head_equal_the_two_calls(gpu_pkg, mode, save_start, t1):
    """lrnde_node_forward_record_ce == lrnde_node_forward_record then lrnde_classifier_ce, bit for bit (forward values, loss,
    logits, both cotangents), and the backward from its record equals the backward from the two calls' record"""
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    P = gpu_pkg
    D, H, K, B = 784, 100, 10, 96
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    p = torch.from_numpy(P.glorot_params(model, seed=3) * np.float32(1.5))
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
    pc = torch.from_numpy((rng.random(K * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).cuda()
    lab = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
    ha, hb = Handle(_mlp_desc(model)), Handle(_mlp_desc(model))
    ha.set_params(p); hb.set_params(p)
    kw = dict(mode=mode, reg_type="error_estimate", t1_or_rand=t1, maxiters=10000, save_start=save_start)
    for rep in range(2):   # (second round: workspaces exist, the head's launches ride ahead of the solve's synchronisation)
        fa = ha.node_forward_record(x, 0.0, 1.0, 1e-5, 1e-5, **kw)
        qa = ha.classifier_ce(fa["u_end"], pc, K, lab)
        ba = ha.node_backward_recorded(qa["du"], w_reg=2.5)
        fb, qb = hb.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, lab, **kw)
        bb = hb.node_backward_recorded(qb["du"], w_reg=2.5)
        assert torch.equal(fa["u_end"], fb["u_end"]) and fa["reg_val"] == fb["reg_val"] and fa["nfe"] == fb["nfe"] and fa["stats"] == fb["stats"]
        assert qa["loss"] == qb["loss"] and torch.equal(qa["logits"], qb["logits"]) and torch.equal(qa["du"], qb["du"]) and torch.equal(qa["dpc"], qb["dpc"])
        assert torch.equal(ba["dx"], bb["dx"]) and torch.equal(ba["dp"], bb["dp"])
    # a failing solve fails the fused call with the solve's status, and the handle goes on
    with pytest.raises(P.LrndeError) as e:
        hb.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, lab, **dict(kw, maxiters=3))
    assert e.value.code == 1
    bad = lab.clone(); bad[5] = K
    with pytest.raises(P.LrndeError):
        hb.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, bad, **kw)
    fb, qb = hb.node_forward_record_ce(x, 0.0, 1.0, 1e-5, 1e-5, pc, K, lab, **kw)
    assert torch.equal(fa["u_end"], fb["u_end"]) and qa["loss"] == qb["loss"] and torch.equal(qa["du"], qb["du"])
