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

# ---- MNIST-SDE shapes (experiments/src/construct.jl:204-205): one Euler-Heun, one Milstein and one four-stage SRI step ----
Ds, Hs, Bs = 32, 64, 16
rngs = np.random.default_rng(3)
pds = (O.glorot_mlp_params(Ds, Hs, time_dep=False, seed=3) + rngs.standard_normal(O.lib().lro_mlp_param_count(Ds, Hs, 0)).astype(np.float32) * np.float32(0.02)).astype(np.float32)
Wg = ((rngs.random((Ds, Ds), dtype=np.float32) - np.float32(0.5)) * np.float32(0.6)).astype(np.float32)
bg = (rngs.standard_normal(Ds) * 0.05).astype(np.float32)
pgs = np.concatenate([Wg.ravel(), bg]).astype(np.float32)
p2 = np.concatenate([np.eye(Ds, dtype=np.float32).ravel(), np.zeros(Ds, np.float32), Wg.ravel(), bg])
drift = O.MlpField(Ds, Hs, pds, time_dep=False, act="tanh", nthreads=4)
diff = O.MlpField(Ds, Ds, p2, time_dep=False, act="identity", nthreads=4)
us = rngs.standard_normal((Bs, Ds)).astype(np.float32)
dts = np.float32(0.05)
dWs = (rngs.standard_normal((Bs, Ds)) * np.sqrt(dts)).astype(np.float32)
dZs = (rngs.standard_normal((Bs, Ds)) * np.sqrt(dts)).astype(np.float32)
tab = rngs.uniform(-0.8, 0.8, len(O.SRI_FIELDS)).astype(np.float32)
eh = O.euler_heun_step(drift, diff, us, dWs, 0.2, dts, 0.14, 0.14, 1.0 / 6.0)
rk = O.rkmil_step(drift, diff, us, dWs, 0.2, dts, 0.14, 0.14)
sr = O.sri_step(drift, diff, dict(zip(O.SRI_FIELDS, tab.tolist())), us, dWs, dZs, 0.2, dts, 0.14, 0.14, 1.0 / 6.0)
np.savez_compressed(os.path.join(HERE, "mnist_sde_b16.npz"), p_drift=pds, p_diffusion=pgs, u=us, dW=dWs, dZ=dZs, t=np.float32(0.2), dt=dts,
                    tableau=tab, eh_u=eh["u"], eh_eest=eh["eest"], eh_reg=eh["reg_val"], rk_u=rk["u"], rk_eest=rk["eest"],
                    rk_reg=rk["reg_val"], sri_u=sr["u"], sri_eest=sr["eest"], sri_reg=sr["reg_val"])
print("wrote", os.path.getsize(os.path.join(HERE, "mnist_sde_b16.npz")), "bytes")
