"""bf16 conv f-eval only (CIFAR shape), for rocprofv3 --kernel-trace --stats runs with LRNDE_CONV_DBG phase switches."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
W, H, B = 32, 32, 256
h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True, compute_dtype=dt)
h.set_params(P.glorot_conv_params(8, 64, seed=0))
u = torch.randn(B, 8, H, W, device="cuda")
us = h.bench_rhs(u, 0.3, reps=10)
print(f"{dt} dbg={os.environ.get('LRNDE_CONV_DBG', 0)}: {us:.1f} us/f-eval", flush=True)
