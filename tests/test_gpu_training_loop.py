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
