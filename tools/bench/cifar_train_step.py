"""run_cifar_training_step at the CIFAR10 shape (B=256, 32x32, tol 1e-4): fwd/bwd split and the stem/head share."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
W = H = 32; B = 256; K = 10
rng = np.random.default_rng(0)
core = P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")), P.Chain(P.Conv((3, 3), 65, 64), P.BatchNorm(64, "gelu")),
                         P.Conv((3, 3), 65, 8)))
node = P.NeuralODE(core, regularize="unbiased", abstol=1e-4, reltol=1e-4, save_start=False, maxiters=10000, compute_dtype=dt) \
    if "compute_dtype" in P.NeuralODE.__init__.__code__.co_varnames else P.NeuralODE(core, regularize="unbiased", abstol=1e-4, reltol=1e-4, save_start=False, maxiters=10000)
pn = P.glorot_conv_params(8, 64, seed=0)
ps = (rng.standard_normal(156) * 0.3).astype(np.float32); ps[140:148] = 1.0; ps[148:156] = 0.0
ph = (rng.standard_normal(73 + K * H * W + K) * 0.05).astype(np.float32)
x = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
params = dict(stem=torch.from_numpy(ps).cuda(), neural_ode=torch.from_numpy(pn).cuda(), head=torch.from_numpy(ph).cuda())
st = node.initialstates(np.random.default_rng(0))
for i in range(2):
    loss, st_, stats, grads, times = P.run_cifar_training_step(node, params, st, x, lab, 2.5)
print(f"loss {loss:.4f} nfe {stats['nfe']} fwd {times['fwd_time']*1e3:.2f} ms bwd {times['bwd_time']*1e3:.2f} ms adjoint {times['adjoint']}")
h = node.handle() if hasattr(node, "handle") else None
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5):
    u0, sb = h.cifar_stem_forward(x, params["stem"], None, return_state=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
for i in range(5):
    hd = h.cifar_head_ce(u0, params["head"], K, lab)
torch.cuda.synchronize(); t2 = time.perf_counter()
for i in range(5):
    h.cifar_stem_backward(x, params["stem"], hd["du"])
torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"stem fwd {(t1-t0)/5*1e3:.3f} ms  head fwd+bwd {(t2-t1)/5*1e3:.3f} ms  stem bwd {(t3-t2)/5*1e3:.3f} ms")
