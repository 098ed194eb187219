# Wall-time breakdown of the training step's backward pass (MNIST-ODE B=512, tol 1.4e-8, CE cotangent):
#   python tools/bench/adjoint_breakdown.py            (LRNDE_ADJ_HOST=1 for the round-1 host-controlled loop)
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, B, tol = 784, 100, 512, 1.4e-8
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
params = P.glorot_params(model, seed=0)
x = torch.from_numpy(np.random.default_rng(0).random((B, D), dtype=np.float32)).cuda()
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(params))
rngc = np.random.default_rng(2)
pc = torch.from_numpy((rngc.random(10 * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(np.sqrt(24.0 / (D + 10)))).cuda()
labels = torch.from_numpy(rngc.integers(0, 10, B).astype(np.int32)).cuda()
def run(w_reg, reps=8):
    tf, tc, tb = [], [], []
    for i in range(reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fw = h.node_forward_record(x, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.3 + 0.05 * i, maxiters=10000)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        head = h.classifier_ce(fw["u_end"], pc, 10, labels)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        bw = h.node_backward_recorded(head["du"], w_reg=w_reg)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if i >= 2: tf.append(t1 - t0); tc.append(t2 - t1); tb.append(t3 - t2)
    return np.mean(tf) * 1e3, np.mean(tc) * 1e3, np.mean(tb) * 1e3, bw["stats_bwd"]
for w in (0.0, 2.5):
    f, c, b, st = run(w)
    print(f"w_reg={w}: forward+record {f:.3f} ms, classifier {c:.3f} ms, backward {b:.3f} ms  (adjoint steps {st['naccept']}+{st['nreject']}, nf {st['nf']})")
