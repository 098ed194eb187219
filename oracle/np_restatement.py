"""Independent numpy restatement of the path (float32 arrays, BLAS matmul, np.tanh) — TEST INFRASTRUCTURE, like
everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
Written from src/perform_step.jl:3-47 and SURVEY.md §3.5, sharing no code with oracle/lrnde_oracle.c.  It serves as
(1) a cross-check of the C oracle and of the HIP path that is not co-designed with the kernels' summation order, and
(2) the "numpy/OpenBLAS restatement" CPU baseline of SURVEY.md §8(d)(2) / BASELINE.md §3.2 (OpenBLAS is also the BLAS
Julia's LinearAlgebra uses for the reference's Dense layers).  "Parity unpinned": no output of the Julia reference exists
in this image to check it against."""
import numpy as np

f32 = np.float32
C = [0.161, 0.327, 0.9, 0.9800255409045097]
A = {2: [0.161],
     3: [-0.008480655492356989, 0.335480655492357],
     4: [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
     5: [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
     6: [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
     7: [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]}
BT = [-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
      0.5823571654525552, -0.45808210592918697, 0.015151515151515152]


def gelu(x):
    x = x.astype(np.float64)
    return (0.5 * x * (1 + np.tanh(np.sqrt(2 / np.pi) * (x + 0.044715 * x ** 3)))).astype(f32)


class NpMlp:
    def __init__(self, D, H, p, time_dep=True, act="tanh"):
        td = int(time_dep)
        self.D, self.H, self.td, self.act = D, H, td, act
        o = 0
        self.W1 = p[o:o + H * (D + td)].reshape(D + td, H).T.copy(); o += H * (D + td)   # (H, D+td)
        self.b1 = p[o:o + H].copy(); o += H
        self.W2 = p[o:o + D * (H + td)].reshape(H + td, D).T.copy(); o += D * (H + td)   # (D, H+td)
        self.b2 = p[o:o + D].copy()

    def __call__(self, u, t):  # u: (B, D)
        B = u.shape[0]
        if self.act not in ("tanh", "identity") or not self.td:
            tcol = np.full((B, 1), t, dtype=f32)
            x = np.concatenate([u, tcol], axis=1) if self.td else u
            h = x @ self.W1.T + self.b1
            h = np.tanh(h) if self.act == "tanh" else (gelu(h) if self.act == "gelu" else h)
            h = np.concatenate([h.astype(f32), tcol], axis=1) if self.td else h.astype(f32)
            return (h @ self.W2.T + self.b2).astype(f32)
        # the MNIST field, without per-call temporaries: [u; t] and [h; t] live in preallocated buffers (the reference's
        # TDChain allocates them on every call, src/layers/common.jl:27-33 — this baseline does not charge that)
        buf = getattr(self, "_buf", None)
        if buf is None or buf[0].shape[0] != B:
            buf = self._buf = (np.empty((B, self.D + 1), f32), np.empty((B, self.H + 1), f32),
                               np.ascontiguousarray(self.W1.T), np.ascontiguousarray(self.W2.T))
        xc, hc, W1T, W2T = buf
        xc[:, :self.D] = u
        xc[:, self.D] = f32(t)
        h = hc[:, :self.H]
        np.matmul(xc, W1T, out=h)
        h += self.b1
        if self.act == "tanh":
            np.tanh(h, out=h)
        hc[:, self.H] = f32(t)
        out = hc @ W2T
        out += self.b2
        return out


class TorchMlp(NpMlp):
    """The same field with torch CPU kernels (MKL sgemm via addmm, vectorised tanh) — bench.py's third CPU leg (VERDICT r2
    item 10: OpenBLAS through numpy reached 152 GFLOP/s on 16 cores; MKL is the other BLAS in the image).  float32,
    zero-copy views of the solver's numpy arrays."""

    def __init__(self, D, H, p, time_dep=True, act="tanh", threads=None):
        super().__init__(D, H, p, time_dep, act)
        import torch
        assert time_dep and act == "tanh", "TorchMlp: the MNIST field only"
        self.torch = torch
        if threads:
            torch.set_num_threads(int(threads))
        self.W1T = torch.from_numpy(np.ascontiguousarray(self.W1.T))
        self.W2T = torch.from_numpy(np.ascontiguousarray(self.W2.T))
        self.b1t, self.b2t = torch.from_numpy(self.b1), torch.from_numpy(self.b2)
        self._tb = None

    def __call__(self, u, t):
        torch = self.torch
        B = u.shape[0]
        if self._tb is None or self._tb[0].shape[0] != B:
            self._tb = (torch.empty((B, self.D + 1)), torch.empty((B, self.H + 1)))
        xc, hc = self._tb
        xc[:, :self.D] = torch.from_numpy(u)
        xc[:, self.D] = float(f32(t))
        h = torch.addmm(self.b1t, xc, self.W1T)
        torch.tanh_(h)
        hc[:, :self.H] = h
        hc[:, self.H] = float(f32(t))
        return torch.addmm(self.b2t, hc, self.W2T).numpy()


def rms(x):
    return f32(np.sqrt(np.mean(np.square(x.astype(np.float64)))))


def rms32(x):
    """sqrt(sum(abs2, x)/length(x)) with a float32 BLAS dot product (what ODE_DEFAULT_NORM's Float32 @fastmath loop
    amounts to)"""
    v = x.reshape(-1)
    return f32(np.sqrt(np.dot(v, v) / f32(v.size)))


def tsit5_step(f, uprev, k1, t, dt, abstol, reltol, fast=False):
    """src/perform_step.jl:10-47.  fast=False: the stage sums term by term, left to right, float64-accumulated norms
    (cross-check use).  fast=True (the CPU-baseline leg of bench.py): each stage sum / utilde as ONE sgemv over the
    stacked k's (the fused loop a Julia broadcast compiles to, here multi-threaded BLAS) and float32 BLAS norms."""
    t, dt = f32(t), f32(dt)
    cs = [C[0], C[1], C[2], C[3], 1.0, 1.0]
    if not fast:
        ks = [k1]
        xs = {}
        for s in range(2, 8):
            acc = sum(f32(a) * k for a, k in zip(A[s], ks))
            x = (uprev + dt * acc).astype(f32)
            xs[s] = x
            ks.append(f(x, f32(t + f32(cs[s - 2]) * dt)))
        u, g6 = xs[7], xs[6]
        utilde = dt * sum(f32(b) * k for b, k in zip(BT, ks))
        resid = utilde / (f32(abstol) + np.maximum(np.abs(uprev), np.abs(u)) * f32(reltol))
        eest = rms(resid)
        den = rms(u - g6)
        stiff = f32(0) if den == 0 else f32(abs(rms(ks[6] - ks[5]) / (den + np.finfo(f32).eps)) / f32(3.5068))
        return dict(u=u, k7=ks[6], eest=eest, reg_error=f32(eest * dt), reg_stiff=stiff, ks=ks)
    n = uprev.size
    K = np.empty((7, n), f32)
    K[0] = k1.reshape(-1)
    up = uprev.reshape(-1)
    xs = {}
    for s in range(2, 8):
        x = np.asarray(A[s], f32) @ K[:s - 1]
        x *= dt
        x += up
        xs[s] = x
        K[s - 1] = f(x.reshape(uprev.shape), f32(t + f32(cs[s - 2]) * dt)).reshape(-1)
    u, g6 = xs[7], xs[6]
    utilde = np.asarray(BT, f32) @ K
    utilde *= dt
    sc = np.maximum(np.abs(up), np.abs(u))
    sc *= f32(reltol)
    sc += f32(abstol)
    utilde /= sc
    eest = rms32(utilde)
    den = rms32(u - g6)
    stiff = f32(0) if den == 0 else f32(abs(rms32(K[6] - K[5]) / (den + np.finfo(f32).eps)) / f32(3.5068))
    ks = [K[i].reshape(uprev.shape) for i in range(7)]
    return dict(u=u.reshape(uprev.shape), k7=ks[6], eest=eest, reg_error=f32(eest * dt), reg_stiff=stiff, ks=ks)


# Tsit5 dense output: u(t + theta*dt) = uprev + dt * sum_i b_i(theta) k_i
_R = [[1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216],
      [0.13169999999999998, -0.2234, 0.1017],
      [3.9302962368947516, -5.941033872131505, 2.490627285651253],
      [-12.411077166933676, 30.33818863028232, -16.548102889244902],
      [37.50931341651104, -88.1789048947664, 47.37952196281928],
      [-27.896526289197286, 65.09189467479366, -34.87065786149661],
      [1.5, -4.0, 2.5]]


def tsit5_interp(uprev, ks, dt, theta):
    th = f32(theta)
    b = [f32(th * f32(f32(_R[0][0]) + th * f32(f32(_R[0][1]) + th * f32(f32(_R[0][2]) + th * f32(_R[0][3])))))]
    for r in _R[1:]:
        b.append(f32(f32(th * th) * f32(f32(r[0]) + th * f32(f32(r[1]) + th * f32(r[2])))))
    acc = sum(bi * k for bi, k in zip(b, ks))
    return (uprev + f32(dt) * acc).astype(f32)


# ---------------------------------------------------------------------------------------------
# float64 field + the adaptive loop, written from SURVEY.md §3.5 (OrdinaryDiffEq's solve for Tsit5, out-of-place,
# Float32 u and t) — shares no code and no summation order with oracle/lrnde_oracle.c or the HIP kernels: the field is
# evaluated in float64 BLAS from the float32 state and rounded once ("the exact fp32 field"), everything the reference
# does in Float32 broadcasts (stage sums, utilde, residual) is done in float32 numpy, norms accumulate in float64.
# Used by tests/test_gpu_independent_parity.py to check the HIP path against something that is NOT the co-designed oracle.
# ---------------------------------------------------------------------------------------------
class NpMlp64(NpMlp):
    def __init__(self, D, H, p, time_dep=True, act="tanh"):
        super().__init__(D, H, p, time_dep, act)
        self.W1d, self.b1d = self.W1.astype(np.float64), self.b1.astype(np.float64)
        self.W2d, self.b2d = self.W2.astype(np.float64), self.b2.astype(np.float64)

    def __call__(self, u, t):
        B = u.shape[0]
        tcol = np.full((B, 1), np.float64(f32(t)))
        x = np.concatenate([u.astype(np.float64), tcol], axis=1) if self.td else u.astype(np.float64)
        pre = x @ self.W1d.T + self.b1d
        if self.act == "tanh":
            h = np.tanh(pre)
        elif self.act == "gelu":
            h = 0.5 * pre * (1 + np.tanh(np.sqrt(2 / np.pi) * (pre + 0.044715 * pre ** 3)))
        else:
            h = pre
        h = np.concatenate([h, tcol], axis=1) if self.td else h
        return (h @ self.W2d.T + self.b2d).astype(f32)


def fastpow(x, y):
    """DiffEqBase.fastpow(::Float32, ::Float32) of the reference's version window, from its description in SURVEY.md §3.5:
    fastpow2(y * fastlog2(x)), integer / float32 operations only."""
    x, y = f32(x), f32(y)
    bits = np.array([x], dtype=f32).view(np.uint32)[0]
    if bits & np.uint32(0x00400000):
        signif = np.array([(bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F000000)], dtype=np.uint32).view(f32)[0]
        fexp = f32(int((bits >> np.uint32(23)) & np.uint32(0xFF)) - 126)
    else:
        signif = np.array([(bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)], dtype=np.uint32).view(f32)[0]
        fexp = f32(int((bits >> np.uint32(23)) & np.uint32(0xFF)) - 127)
    s = f32(signif - f32(1))
    lg2 = f32(fexp + f32(f32(s * f32(f32(f32(0.338953) * s) + f32(2.198599))) / f32(s + f32(1.523692))))
    z_in = f32(y * lg2)
    offset = f32(1) if z_in < 0 else f32(0)
    clipp = max(z_in, f32(-126))
    w = f32(np.trunc(clipp))
    z = f32(f32(clipp - w) + offset)
    v = f32(f32(f32(clipp + f32(121.2740575)) + f32(f32(27.7280233) / f32(f32(4.84252568) - z))) - f32(f32(1.49012907) * z))
    ib = np.uint32(int(f32(f32(1 << 23) * v)))
    return np.array([ib], dtype=np.uint32).view(f32)[0]


def _eps(x):
    return np.spacing(f32(abs(x)))


def init_dt(f, u0, t0, t1, abstol, reltol):
    t0, t1, abstol, reltol = f32(t0), f32(t1), f32(abstol), f32(reltol)
    sk = abstol + np.abs(u0) * reltol
    f0 = f(u0, t0)
    d0, d1 = rms(u0 / sk), rms(f0 / sk)
    dtmax = f32(t1 - t0)
    dt0 = f32(1e-6) if (float(d0) < 1e-5 or float(d1) < 1e-5) else f32(f32(d0 / d1) / f32(100))
    dt0 = min(dt0, dtmax)
    f1 = f((u0 + dt0 * f0).astype(f32), f32(t0 + dt0))
    d2 = f32(rms((f1 - f0) / sk) / dt0)
    md = max(d1, d2)
    if float(md) <= 1e-15:
        dt1 = max(f32(1e-6), f32(dt0 * f32(1e-3)))
    else:
        dt1 = f32(10.0 ** float(f32(-(f32(2) + f32(np.log10(float(md)))) / f32(5))))
    return min(f32(f32(100) * dt0), dt1, dtmax), f0


def solve(f, u0, t0, t1, abstol, reltol, maxiters=10000, save_t=None, fast=False):
    """adaptive Tsit5 from t0 to t1; returns dict(u, naccept, nreject, nf, dts[, u_save = sol(save_t) by dense output])"""
    u_save = None
    t0, t1 = f32(t0), f32(t1)
    dt, k1 = init_dt(f, u0, t0, t1, abstol, reltol)
    gamma, qmin, qmax, qoldinit = f32(0.9), f32(0.2), f32(10), f32(1e-4)
    beta1, beta2 = f32(7.0 / 50.0), f32(2.0 / 25.0)
    dtmax = f32(t1 - t0)
    dtmin = max(_eps(t1), _eps(t0))
    t, uprev = t0, u0
    qold, q11, dtpropose = qoldinit, f32(1), dt
    accept, it, naccept, nreject, nf = False, 0, 0, 0, 3
    u = k7 = None
    dts = []
    while t < t1:
        if it > 0:
            if accept:
                uprev, k1, dt = u, k7, dtpropose
            else:
                dt = f32(dt / min(f32(f32(1) / qmin), f32(q11 / gamma)))
        it += 1
        dt = max(min(dtmax, dt), dtmin)
        dt = min(f32(abs(dt)), f32(abs(t1 - t)))
        if it > maxiters or not dt > dtmin:
            raise RuntimeError("restatement solve did not finish")
        r = tsit5_step(f, uprev, k1, t, dt, abstol, reltol, fast=fast)
        u, k7, eest = r["u"], r["k7"], r["eest"]
        nf += 6
        dts.append(dt)
        if eest == 0:
            q = f32(f32(1) / qmax)
        else:
            q11 = fastpow(eest, beta1)
            q = f32(q11 / fastpow(qold, beta2))
            q = max(f32(f32(1) / qmax), min(f32(f32(1) / qmin), f32(q / gamma)))
        accept = bool(eest <= 1)
        if accept:
            naccept += 1
            dtnew = f32(dt / q)
            qold = max(eest, qoldinit)
            ttmp = f32(t + dt)
            tprev = t
            t = t1 if abs(f32(ttmp - t1)) < f32(f32(100) * _eps(max(t, t1))) else ttmp
            if save_t is not None and u_save is None and f32(save_t) <= t:
                u_save = u if f32(save_t) == t else tsit5_interp(uprev, r["ks"], dt, f32(f32(f32(save_t) - tprev) / dt))
            dtpropose = max(min(dtmax, dtnew), max(_eps(t), dtmin))
        else:
            nreject += 1
    return dict(u=u, naccept=naccept, nreject=nreject, nf=nf, dts=np.array(dts, dtype=f32), u_save=u_save)


def node_forward(f, x, t0, t2, abstol, reltol, t1, fast=False):
    """`(n::NeuralODE{:unbiased, :error_estimate})(x, ps, st)` (src/layers/neural_ode.jl:68-84): solve with saveat
    [t1, t2], fresh init at (sol(t1), t1), one local Tsit5 step; nfe = sol.destats.nf + 6 + 3"""
    sol = solve(f, x, t0, t2, abstol, reltol, save_t=t1, fast=fast)
    dt, k1 = init_dt(f, sol["u_save"], t1, t2, abstol, reltol)
    st = tsit5_step(f, sol["u_save"], k1, t1, dt, abstol, reltol, fast=fast)
    return dict(u_end=sol["u"], reg_val=st["reg_error"], nfe=sol["nf"] + 9, naccept=sol["naccept"], nreject=sol["nreject"])
