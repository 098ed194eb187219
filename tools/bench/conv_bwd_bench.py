import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
W = H = 32; B = 256
h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=True)
h.set_params(P.glorot_conv_params(8, 64, seed=0))
u = torch.randn(B, 8, H, W, device="cuda"); lam = torch.randn_like(u)
for i in range(3): h.vjp(u, 0.3, lam)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): h.vjp(u, 0.3, lam)
torch.cuda.synchronize()
print(f"conv vjp {(time.perf_counter()-t0)/10*1e6:.0f} us", flush=True)
