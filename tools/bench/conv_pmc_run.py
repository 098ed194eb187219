# one CIFAR-shape conv f-eval sequence for the PMC passes (few launches: counters serialise kernels)
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
h = P.ConvHandle(32, 32, 8, 64, act="gelu", bn_train=(os.environ.get("LRNDE_PMC_BN_EVAL") is None), compute_dtype=dt)
h.set_params(P.glorot_conv_params(8, 64, seed=0))
u = torch.randn(256, 8, 32, 32, device="cuda")
if os.environ.get("LRNDE_PMC_VJP"):  # the backward kernels instead: three VJPs
    lam = torch.randn_like(u)
    for i in range(3):
        h.vjp(u, 0.3, lam)
else:
    for i in range(6):
        h.rhs(u, 0.3)
torch.cuda.synchronize()
