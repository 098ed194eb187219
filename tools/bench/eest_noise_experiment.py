import sys, numpy as np, torch
sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import oracle as O
torch.set_num_threads(8)
W = H = 16; B = 2; C, Hc = 8, 64
rng = np.random.default_rng(5)
p = (O.glorot_conv_params(C, Hc, seed=5) * np.float32(1.5)).astype(np.float32)
u0 = rng.standard_normal((B, C * H * W)).astype(np.float32)

def split22(x, scale):  # hi + lo fp16 pair of x*scale, returned as float64 value / scale
    v = (x * scale).to(torch.float32)
    hi = v.to(torch.float16).to(torch.float32)
    lo = (v - hi).to(torch.float16).to(torch.float32)
    return (hi.double() + lo.double()) / scale

def make_field(round_act, round_w, f32_acc=True):
    pt = torch.from_numpy(p.astype(np.float64))
    def f(u, t):
        x = torch.from_numpy(np.asarray(u, dtype=np.float64)).reshape(-1, C, H, W)
        Bn = x.shape[0]
        off = [0]
        def take(n):
            v = pt[off[0]:off[0] + n]; off[0] += n; return v
        def wgt(ci, co, rnd):
            w = take(9 * ci * co).reshape(co, ci, 3, 3)
            if rnd:
                wr = split22(w[:, :ci - 1].float(), 256.0)
                w = torch.cat([wr, w[:, ci - 1:]], dim=1)
            return torch.flip(w, dims=(2, 3))
        tc = lambda z: torch.cat([z, torch.full((Bn, 1, H, W), float(t), dtype=torch.float64)], dim=1)
        gelu = lambda z: 0.5 * z * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))
        def bn(z):
            g, b = take(Hc), take(Hc)
            mu = z.mean(dim=(0, 2, 3), keepdim=True); var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            return gelu((z - mu) / torch.sqrt(var + 1e-5) * g.reshape(1, Hc, 1, 1) + b.reshape(1, Hc, 1, 1))
        rf = (lambda z: z.float().double()) if f32_acc else (lambda z: z)   # results held in fp32 between layers
        z = rf(torch.nn.functional.conv2d(tc(x), wgt(C + 1, Hc, False), padding=1))
        hcur = rf(bn(z))
        if round_act: hcur = split22(hcur.float(), 256.0)
        z = rf(torch.nn.functional.conv2d(tc(hcur), wgt(Hc + 1, Hc, round_w), padding=1))
        hcur = rf(bn(z))
        if round_act: hcur = split22(hcur.float(), 256.0)
        z = rf(torch.nn.functional.conv2d(tc(hcur), wgt(Hc + 1, C, round_w), padding=1))
        return z.reshape(Bn, -1).numpy().astype(np.float32)
    return O.PyField(C * H * W, f)

ref = O.ConvField(W, H, C, Hc, p, nthreads=8)
dt0, k1 = O.init_dt(ref, u0, 0.0, 1.0, 1e-4, 1e-4)
base = O.tsit5_step(ref, u0, k1, 0.0, dt0, 1e-4, 1e-4)["eest"]
print("oracle (C, fp32 chains) first-step EEst", base)
for name, ra, rw in [("float64 field, fp32 between layers", False, False), ("weights 22-bit", False, True), ("activations 22-bit", True, False), ("both 22-bit", True, True)]:
    fld = make_field(ra, rw)
    k1f = fld_rhs = None
    import ctypes
    # k1 from this field
    kk = np.empty_like(u0)
    fld._cb(None, u0.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_float(0.0), B, kk.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    e = O.tsit5_step(fld, u0, kk, 0.0, dt0, 1e-4, 1e-4)["eest"]
    print(f"{name:40s} EEst {e:.6g}  rel dev from oracle {abs(e-base)/base:.3%}")
