"""Generates tests/golden/*.npz from the CPU oracle (the reference is Julia and cannot run
here, so these are regression pins of the restatement, not outputs of the reference)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

D, H, B, tol = 784, 100, 16, 1e-5
p = O.glorot_mlp_params(D, H, seed=0)
x = np.random.default_rng(0).random((B, D), dtype=np.float32)
f = O.MlpField(D, H, p, nthreads=4)
k1 = f.rhs(x, 0.1)
st = O.tsit5_step(f, x, k1, 0.1, 0.05, tol, tol)
sv = O.solve(f, x, 0.0, 1.0, tol, tol, saveat=[0.5, 1.0], maxiters=10000)
nd = O.node_forward(f, x, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.37, maxiters=10000)
np.savez_compressed(os.path.join(HERE, "mnist_mlp_b16.npz"), params=p.astype(np.float16).astype(np.float32) * 0 + p,
                    x=x, k1=k1, t=np.float32(0.1), dt=np.float32(0.05), tol=np.float32(tol),
                    step_u=st["u"], step_k7=st["k7"], step_eest=st["eest"], step_reg_error=st["reg_error"],
                    step_reg_stiff=st["reg_stiff"], solve_dt_trace=sv["trace"]["dt"], solve_u=sv["u"],
                    solve_nf=sv["stats"]["nf"], node_reg_val=nd["reg_val"], node_nfe=nd["nfe"])
print("wrote", os.path.getsize(os.path.join(HERE, "mnist_mlp_b16.npz")), "bytes")

# ---- conv vector field (experiments/src/construct.jl:213-218): small image, the CIFAR block's channel counts ----
Wc, Hh, Cc, Hc, Bc = 8, 8, 8, 64, 2
pc = O.glorot_conv_params(Cc, Hc, seed=0)
rngc = np.random.default_rng(0)
n1 = 9 * (Cc + 1) * Hc
pc[n1:n1 + Hc] = rngc.uniform(0.5, 1.5, Hc); pc[n1 + Hc:n1 + 2 * Hc] = rngc.uniform(-0.3, 0.3, Hc)
xc = rngc.standard_normal((Bc, Wc * Hh * Cc)).astype(np.float32)
lamc = rngc.standard_normal(xc.shape).astype(np.float32)
fc = O.ConvField(Wc, Hh, Cc, Hc, pc, act="gelu", bn_train=True, nthreads=4)
duc = fc.rhs(xc, 0.3)
dyc, gpc = O.conv_vjp(fc, xc, 0.3, lamc)
ndc = O.node_forward(fc, xc, 0.0, 1.0, 1e-3, 1e-3, mode="unbiased", t1_or_rand=0.41)
np.savez_compressed(os.path.join(HERE, "conv_block_8x8_b2.npz"), params=pc, x=xc, lam=lamc, t=np.float32(0.3), du=duc, vjp_dy=dyc,
                    vjp_gp=gpc, node_u_end=ndc["u_end"], node_nfe=ndc["nfe"], node_naccept=ndc["stats"]["naccept"],
                    node_reg_val=ndc["reg_val"])
print("wrote", os.path.getsize(os.path.join(HERE, "conv_block_8x8_b2.npz")), "bytes")
