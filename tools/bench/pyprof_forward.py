# host-side profile of the layer forward's Python wrapper (cProfile over 300 passes)
import cProfile, pstats, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, B = 784, 100, 512
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(P.glorot_params(model, seed=0)))
x = torch.from_numpy(np.random.default_rng(0).random((B, D), dtype=np.float32)).cuda()
def run(n):
    for i in range(n): h.node_forward(x, 0.0, 1.0, 1.4e-8, 1.4e-8, mode="unbiased", t1_or_rand=0.3, maxiters=10000)
run(20)
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
