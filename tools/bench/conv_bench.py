import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
shapes = [(32, 32, 256)] if os.environ.get("LRNDE_CONV_DBG") else [(32, 32, 256), (28, 28, 512), (32, 32, 32)]
for dt in ("f32", "bf16"):
    for (W, H, B) in shapes:
        h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True, compute_dtype=dt)
        h.set_params(P.glorot_conv_params(8, 64, seed=0))
        u = torch.randn(B, 8, H, W, device="cuda")
        us = h.bench_rhs(u, 0.3, reps=10)
        fl = 2 * W * H * (81 * 64 + 585 * 64 + 585 * 8) * B
        print(f"{dt} dbg={os.environ.get('LRNDE_CONV_DBG', 0)} W={W} H={H} B={B}: {us:.1f} us/f-eval  {fl / us / 1e6:.2f} TFLOP/s", flush=True)
