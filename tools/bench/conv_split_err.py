# accuracy of the conv f-eval and of one step's EEst against the oracle: f32 split path vs native fp32 MFMA path
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lrnde_amd as P, oracle as O
W = H = 16; B = 2; C, Hc = 8, 64
rng = np.random.default_rng(5)
p = (O.glorot_conv_params(C, Hc, seed=5) * np.float32(1.5)).astype(np.float32)
u = rng.standard_normal((B, C, H, W)).astype(np.float32)
fld = O.ConvField(W, H, C, Hc, p, nthreads=8)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
h = P.ConvHandle(W, H, C, Hc, compute_dtype=dtype); h.set_params(p)
ud = torch.from_numpy(u).cuda()
ref = fld.rhs(u.reshape(B, -1), 0.3).reshape(u.shape)
got = h.rhs(ud, 0.3).cpu().numpy()
sc = np.abs(ref).max()
print("dtype", dtype, " rhs max err / scale %.3e  rms err / rms %.3e" % (np.abs(got - ref).max() / sc, np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))))
dt0, k1 = O.init_dt(fld, u.reshape(B, -1), 0.0, 1.0, 1e-4, 1e-4)
so = O.tsit5_step(fld, u.reshape(B, -1), k1, 0.0, dt0, 1e-4, 1e-4)
dtg, k1g = h.init_dt(ud, 0.0, 1.0, 1e-4, 1e-4)
sg = h.perform_step(ud, k1g, 0.0, float(dt0), 1e-4, 1e-4)
print("  init dt oracle %.6g gpu %.6g ; first-step EEst oracle %.6g gpu %.6g" % (dt0, dtg, so["eest"], sg["eest"]))
for dt in (0.1, 0.2):
    so = O.tsit5_step(fld, u.reshape(B, -1), k1, 0.0, dt, 1e-4, 1e-4); sg = h.perform_step(ud, k1g, 0.0, dt, 1e-4, 1e-4)
    print("  dt %.2f EEst oracle %.6g gpu %.6g  rel diff %.2e" % (dt, so["eest"], sg["eest"], abs(sg["eest"] - so["eest"]) / so["eest"]))
