# Soak / end-to-end check: a few hundred training steps of the MNIST-ODE model (run_training_step + Adam update of the NeuralODE
# and classifier parameters) on a fixed synthetic classification task, twice — companion stream on and off — from the same
# start.  The loss must fall, every step must succeed, and the two runs must end with the same parameters, bit for bit.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
D, H, K, B = 784, 100, 10, 128
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
rng = np.random.default_rng(0)
centers = rng.random((K, D), dtype=np.float32)
def batch(i):
    g = np.random.default_rng(1000 + i)
    lab = g.integers(0, K, B).astype(np.int32)
    x = (centers[lab] + 0.15 * g.standard_normal((B, D)).astype(np.float32)).clip(0, 1).astype(np.float32)
    return torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()
def run(overlap):
    node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=1e-4, reltol=1e-4, save_start=False, maxiters=10000)
    ps = torch.from_numpy(P.glorot_params(model, seed=0)).cuda()
    pc = torch.from_numpy((np.random.default_rng(2).random(K * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).cuda()
    opt = P.Optimiser("adam", learning_rate=1e-3)
    st = node.initialstates(np.random.default_rng(3))
    node._bind(ps, None).set_overlap(overlap)
    losses = []
    for i in range(NSTEP):
        x, lab = batch(i)
        loss, st, stats, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
        opt.update([ps, pc], [grads["neural_ode"], grads["classifier"]])
        losses.append(float(loss))
        if not np.isfinite(losses[-1]): raise SystemExit(f"non-finite loss at step {i}")
    return ps.clone(), pc.clone(), losses
t0 = time.time()
pa, ca, la = run(True)
pb, cb, lb = run(False)
same = torch.equal(pa, pb) and torch.equal(ca, cb) and la == lb
print(f"{NSTEP} steps x 2 in {time.time() - t0:.0f} s: loss {la[0]:.4f} -> {np.mean(la[-10:]):.4f}; companion stream on/off identical: {same}", flush=True)
sys.exit(0 if same and np.mean(la[-10:]) < 0.7 * la[0] else 1)
