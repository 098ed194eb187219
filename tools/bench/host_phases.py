# Where the host spends a layer forward (lrnde_host_phases) next to the Python-side wall clock per call:
#   python tools/bench/host_phases.py [passes]
import ctypes as C, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd import _lib as L
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, B = 784, 100, 512
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(P.glorot_params(model, seed=0)))
x = torch.from_numpy(np.random.default_rng(0).random((B, D), dtype=np.float32)).cuda()
t1s = np.random.default_rng(1).random(N + 20, dtype=np.float32)
def run(i): return h.node_forward(x, 0.0, 1.0, 1.4e-8, 1.4e-8, mode="unbiased", t1_or_rand=float(t1s[i]), maxiters=10000)
for i in range(20): run(i)
us = (C.c_double * 4)()
L.lib.lrnde_host_phases(h._ctx, us, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): run(20 + i)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / N * 1e6
L.lib.lrnde_host_phases(h._ctx, us, 1)
inside = sum(us)
print(f"per call: wall {el:.1f} us; inside lrnde_node_forward {inside:.1f} us = enqueue init {us[0]:.1f} + feed loop {us[1]:.1f} + "
      f"final sync {us[2]:.1f} + exit {us[3]:.1f}; outside the library (ctypes, tensors, dict) {el - inside:.1f} us")
