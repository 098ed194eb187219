import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import lrnde_amd as P
for (W, H, B) in [(32, 32, 256), (28, 28, 512), (32, 32, 32)]:
    h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True)
    h.set_params(P.glorot_conv_params(8, 64, seed=0))
    u = torch.randn(B, 8, H, W, device="cuda")
    us = h.bench_rhs(u, 0.3, reps=10)
    fl = 2 * W * H * (81 * 64 + 585 * 64 + 585 * 8) * B
    print(f"W={W} H={H} B={B}: {us:.1f} us/f-eval  {fl / us / 1e6:.2f} TFLOP/s", flush=True)
