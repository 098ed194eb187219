# One NeuralDSDE layer forward at BASELINE config 5 repeated a few times, for an API/kernel timeline:
#   rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/sdetrace -- python3 tools/bench/sde_layer_trace.py
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import _mlp_desc
D, H, B, nfine = 32, 64, 512, 256
f32 = np.float32
rng = np.random.default_rng(0)
h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
h.set_params((rng.standard_normal(D * H + H + H * D + D) * 0.3).astype(f32), (rng.standard_normal(D * D + D) * 0.05).astype(f32))
x = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
z = torch.from_numpy(rng.standard_normal((B, D)).astype(f32)).cuda()
hh = f32(1.0 / nfine)
W = torch.from_numpy(np.concatenate([np.zeros((1, B, D), f32), np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(hh)).astype(f32), axis=0, dtype=f32)], axis=0)).cuda()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.cuda.synchronize()
    fw = h.node_forward_record(x, W, 0.0, 1.0, 0.14, 0.14, z_local=z, mode="unbiased", t1_or_rand=0.4, saveat=(), save_start=-1)
    torch.cuda.synchronize()
print(fw["stats"])
