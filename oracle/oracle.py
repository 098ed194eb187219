"""ctypes binding of the CPU oracle (oracle/lrnde_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.
Arrays are numpy float32, column-major (D x B) states passed as C-contiguous
(B, D) arrays (sample-major), which is the same memory.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblrnde_oracle.so")

ACT = {"identity": 0, "tanh": 1, "gelu": 2}
REG = {"error_estimate": 0, "stiffness_estimate": 1}
MODE = {"none": 0, "unbiased": 1, "biased": 2}
RETCODES = {0: "Success", 1: "MaxIters", 2: "DtLessThanMin", 3: "DtNaN", 4: "BadArg", 5: "Capacity"}

FIELD_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_int, C.POINTER(C.c_float))


class Field(C.Structure):
    _fields_ = [("fn", FIELD_FN), ("ctx", C.c_void_p), ("D", C.c_int)]


class Mlp(C.Structure):
    _fields_ = [("D", C.c_int), ("H", C.c_int), ("time_dep", C.c_int), ("act", C.c_int),
                ("p", C.POINTER(C.c_float)), ("nthreads", C.c_int)]


class Conv(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("C", C.c_int), ("Hc", C.c_int), ("act", C.c_int),
                ("bn_train", C.c_int), ("eps", C.c_float), ("p", C.POINTER(C.c_float)),
                ("bn_state", C.POINTER(C.c_float)), ("nthreads", C.c_int), ("bn_run", C.POINTER(C.c_float)),
                ("bf16", C.c_int)]


class Opts(C.Structure):
    _fields_ = [("abstol", C.c_float), ("reltol", C.c_float), ("maxiters", C.c_int),
                ("save_start", C.c_int), ("save_everystep", C.c_int), ("exact_pow", C.c_int), ("alg", C.c_int)]


ALGS = {"tsit5": 0, "vcab3": 1, "vcabm3": 2}


class Stats(C.Structure):
    _fields_ = [("retcode", C.c_int), ("nf", C.c_int), ("naccept", C.c_int), ("nreject", C.c_int),
                ("iters", C.c_int), ("nsaved", C.c_int), ("t_final", C.c_float),
                ("dt_final", C.c_float), ("eest_last", C.c_float), ("dt_init", C.c_float)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TraceRow(C.Structure):
    _fields_ = [("t", C.c_float), ("dt", C.c_float), ("eest", C.c_float), ("accepted", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "lrnde_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        for name in ("lro_expf", "lro_tanhf", "lro_geluf", "lro_fastlog2", "lro_fastpow2"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.lro_fastpow.restype = C.c_float
        L.lro_fastpow.argtypes = [C.c_float, C.c_float]
        L.lro_mlp_param_count.argtypes = [C.c_int] * 3
        L.lro_mlp_rhs.restype = None
        L.lro_mlp_rhs.argtypes = [C.POINTER(Mlp), fp, C.c_float, C.c_int, fp]
        L.lro_classifier_ce.restype = C.c_float
        L.lro_classifier_ce.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_int), fp, fp, fp]
        ip = C.POINTER(C.c_int)
        L.lro_cifar_stem_forward.restype = None
        L.lro_cifar_stem_forward.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_float, fp, fp]
        L.lro_cifar_stem_backward.restype = None
        L.lro_cifar_stem_backward.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_float, fp, fp]
        L.lro_cifar_head_ce.restype = C.c_float
        L.lro_cifar_head_ce.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, ip, fp, fp, fp]
        L.lro_cifar_head_param_count.argtypes = [C.c_int] * 3
        L.lro_conv_param_count.argtypes = [C.c_int] * 2
        L.lro_conv_rhs.restype = None
        L.lro_conv_rhs.argtypes = [C.POINTER(Conv), fp, C.c_float, C.c_int, fp]
        L.lro_conv_vjp.restype = None
        L.lro_conv_vjp.argtypes = [C.POINTER(Conv), fp, C.c_float, fp, C.c_int, fp, fp]
        L.lro_conv_step_reg_grad.argtypes = [C.POINTER(Conv), fp, fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.c_int, C.c_int, fp, C.POINTER(C.c_float)]
        L.lro_conv_node_backward.argtypes = [C.POINTER(Conv), fp, C.c_int, C.c_float, C.c_float, C.POINTER(Opts), C.c_int,
                                             C.c_int, C.c_float, fp, C.c_float, fp, fp, C.POINTER(Stats), C.POINTER(Stats)]
        L.lro_conv_as_field.restype = None
        L.lro_conv_as_field.argtypes = [C.POINTER(Conv), C.POINTER(Field)]
        L.lro_mlp_as_field.restype = None
        L.lro_mlp_as_field.argtypes = [C.POINTER(Mlp), C.POINTER(Field)]
        L.lro_tsit5_step.argtypes = [C.POINTER(Field), fp, fp, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int, fp, fp, fp, fp, fp, fp, fp]
        L.lro_tsit5_step_sums.argtypes = [C.POINTER(Field), fp, fp, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_int, fp, fp, fp, fp, C.POINTER(C.c_double)]
        L.lro_init_dt.argtypes = [C.POINTER(Field), fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int, fp, fp]
        L.lro_tsit5_interp.restype = None
        L.lro_tsit5_interp.argtypes = [C.c_float, C.c_float, fp, C.POINTER(fp), C.c_long, fp]
        L.lro_solve.argtypes = [C.POINTER(Field), fp, C.c_int, C.c_float, C.c_float, C.POINTER(Opts),
                                fp, C.c_int, fp, fp, C.c_int, C.POINTER(Stats), C.POINTER(TraceRow),
                                C.c_int]
        L.lro_node_forward.argtypes = [C.POINTER(Field), fp, C.c_int, C.c_float, C.c_float,
                                       C.POINTER(Opts), C.c_int, C.c_int, C.c_float, fp, fp,
                                       C.POINTER(C.c_int), C.POINTER(Stats), fp]
        L.lro_euler_heun_step.argtypes = [C.POINTER(Field), C.POINTER(Field), fp, fp, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp, fp,
                                          fp]
        L.lro_rkmil_step.argtypes = [C.POINTER(Field), C.POINTER(Field), fp, fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_int, fp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.lro_tsit5_tableau.argtypes = [C.POINTER(C.c_double)] * 4
        L.lro_mlp_vjp.restype = None
        L.lro_mlp_vjp.argtypes = [C.POINTER(Mlp), fp, C.c_float, fp, C.c_int, fp, fp]
        L.lro_tsit5_step_reg_grad.argtypes = [C.POINTER(Mlp), fp, fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                              C.c_int, C.c_int, fp, fp]
        L.lro_node_backward.argtypes = [C.POINTER(Mlp), fp, C.c_int, C.c_float, C.c_float, C.POINTER(Opts), C.c_int,
                                        C.c_int, C.c_float, fp, C.c_float, fp, fp, C.POINTER(Stats), C.POINTER(Stats)]
        L.lro_node_backward_traced.argtypes = L.lro_node_backward.argtypes + [C.POINTER(TraceRow), C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def vec_fn(name, x):
    f = getattr(lib(), name)
    x = np.asarray(x, dtype=np.float32)
    return np.array([f(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def tableau():
    a = (C.c_double * 21)(); c = (C.c_double * 6)(); bt = (C.c_double * 7)(); r = (C.c_double * 28)()
    lib().lro_tsit5_tableau(a, c, bt, r)
    return np.array(a), np.array(c), np.array(bt), np.array(r).reshape(7, 4)


class MlpField:
    """TDChain(Dense(D+td->H, act), Dense(H+td->D)) with flat Lux-ordered params."""

    def __init__(self, D, H, params, time_dep=True, act="tanh", nthreads=1):
        self.D, self.H = int(D), int(H)
        self.params = _f32(params)
        assert self.params.size == lib().lro_mlp_param_count(D, H, int(time_dep)), "param count"
        self.m = Mlp(D, H, int(time_dep), ACT[act], _fp(self.params), int(nthreads))
        self.field = Field()
        lib().lro_mlp_as_field(C.byref(self.m), C.byref(self.field))

    def rhs(self, u, t):
        u = _f32(u)
        B = u.size // self.D
        du = np.empty_like(u)
        lib().lro_mlp_rhs(C.byref(self.m), _fp(u), float(t), B, _fp(du))
        return du


class ConvField:
    """TDChain(Chain(Conv3x3(C+1=>Hc), BN(Hc,act)), Chain(Conv(Hc+1=>Hc), BN(Hc,act)), Conv(Hc+1=>C))
    on a (W,H,C) image state (experiments/src/construct.jl:213-218), flat Lux-ordered params."""

    def __init__(self, W, H, Cch, Hc, params, act="gelu", bn_train=True, bn_state=None, eps=1e-5, nthreads=1,
                 bf16=False):
        self.W, self.H, self.C, self.Hc = int(W), int(H), int(Cch), int(Hc)
        self.D = self.W * self.H * self.C
        self.params = _f32(params)
        assert self.params.size == lib().lro_conv_param_count(self.C, self.Hc), "param count"
        self.bn_state = None if bn_state is None else _f32(bn_state)
        self.bn_run = np.concatenate([np.zeros(Hc), np.ones(Hc), np.zeros(Hc), np.ones(Hc)]).astype(np.float32)
        self.m = Conv(self.W, self.H, self.C, self.Hc, ACT[act], int(bool(bn_train)), float(eps), _fp(self.params),
                      _fp(self.bn_state) if self.bn_state is not None else None, int(nthreads), _fp(self.bn_run),
                      int(bool(bf16)))
        self.field = Field()
        lib().lro_conv_as_field(C.byref(self.m), C.byref(self.field))

    def rhs(self, u, t):
        u = _f32(u)
        B = u.size // self.D
        du = np.empty_like(u)
        lib().lro_conv_rhs(C.byref(self.m), _fp(u), float(t), B, _fp(du))
        return du


def conv_vjp(fld, y, t, lam, want_gp=True):
    y = _f32(y); lam = _f32(lam)
    B = y.size // fld.D
    dy = np.empty_like(y)
    gp = np.zeros(fld.params.size, np.float32) if want_gp else None
    lib().lro_conv_vjp(C.byref(fld.m), _fp(y), float(t), _fp(lam), B, _fp(dy), _fp(gp))
    return dy, gp


def glorot_conv_params(Cch, Hc, seed=0):
    """Lux Conv init: weight ~ glorot_uniform over (3,3,cin,cout) (fan_in = 9 cin, fan_out = 9 cout), no
    bias; BatchNorm scale = 1, bias = 0.  numpy stream (the Julia RNG streams cannot be reproduced)."""
    rng = np.random.default_rng(seed)
    def glorot(cin, cout):
        return ((rng.random(9 * cin * cout, dtype=np.float32) - np.float32(0.5)) *
                np.float32(np.sqrt(24.0 / (9 * cin + 9 * cout)))).astype(np.float32)
    one, zero = np.ones(Hc, np.float32), np.zeros(Hc, np.float32)
    return np.concatenate([glorot(Cch + 1, Hc), one, zero, glorot(Hc + 1, Hc), one, zero, glorot(Hc + 1, Cch)])


class PyField:
    """Arbitrary python vector field f(u[(B,D)], t) -> du, for known-answer tests."""

    def __init__(self, D, fn):
        self.D = int(D)

        def tramp(_ctx, up, t, B, dup):
            u = np.ctypeslib.as_array(up, shape=(B, self.D))
            du = np.ctypeslib.as_array(dup, shape=(B, self.D))
            du[...] = np.asarray(fn(u.copy(), np.float32(t)), dtype=np.float32)

        self._cb = FIELD_FN(tramp)
        self.field = Field(self._cb, None, self.D)


def make_opts(abstol, reltol, maxiters=1000, save_start=False, save_everystep=False, exact_pow=False, solver="tsit5"):
    return Opts(float(abstol), float(reltol), int(maxiters), int(save_start), int(save_everystep),
                int(exact_pow), ALGS[solver])


def tsit5_step(fld, uprev, k1, t, dt, abstol, reltol, want_stages=False):
    uprev = _f32(uprev); k1 = _f32(k1)
    D = fld.D; B = uprev.size // D
    u = np.empty_like(uprev); k7 = np.empty_like(uprev)
    ks = np.empty((5,) + uprev.shape, dtype=np.float32)
    g6 = np.empty_like(uprev)
    ee = C.c_float(); re = C.c_float(); rs = C.c_float()
    rc = lib().lro_tsit5_step(C.byref(fld.field), _fp(uprev), _fp(k1), float(t), float(dt),
                              float(abstol), float(reltol), B, _fp(u), _fp(k7), _fp(ks), _fp(g6),
                              C.byref(ee), C.byref(re), C.byref(rs))
    assert rc == 0
    out = dict(u=u, k7=k7, eest=np.float32(ee.value), reg_error=np.float32(re.value),
               reg_stiff=np.float32(rs.value))
    if want_stages:
        out["ks"] = ks
        out["g6"] = g6
    return out


def tsit5_step_sums(fld, uprev, k1, t, dt, abstol, reltol):
    """One step on this array; returns u, k7 and the raw fp64 sums (for batch-sharded callers)."""
    uprev = _f32(uprev); k1 = _f32(k1)
    B = uprev.size // fld.D
    u = np.empty_like(uprev); k7 = np.empty_like(uprev)
    ks = np.empty((5,) + uprev.shape, dtype=np.float32)
    sums = (C.c_double * 3)()
    rc = lib().lro_tsit5_step_sums(C.byref(fld.field), _fp(uprev), _fp(k1), float(t), float(dt),
                                   float(abstol), float(reltol), B, _fp(u), _fp(k7), _fp(ks), None, sums)
    assert rc == 0
    return dict(u=u, k7=k7, ks=ks, sums=np.array(sums, dtype=np.float64))


def init_dt(fld, u0, t0, tend, abstol, reltol):
    u0 = _f32(u0)
    B = u0.size // fld.D
    f0 = np.empty_like(u0)
    dt = C.c_float()
    lib().lro_init_dt(C.byref(fld.field), _fp(u0), float(t0), float(tend), float(abstol),
                      float(reltol), B, _fp(f0), C.byref(dt))
    return np.float32(dt.value), f0


def interp(theta, dt, y0, ks7):
    y0 = _f32(y0)
    arrs = [_f32(k) for k in ks7]
    ptrs = (C.POINTER(C.c_float) * 7)(*[_fp(a) for a in arrs])
    out = np.empty_like(y0)
    lib().lro_tsit5_interp(float(theta), float(dt), _fp(y0), ptrs, y0.size, _fp(out))
    return out


def solve(fld, u0, t0, t1, abstol, reltol, saveat=(), maxiters=1000, save_start=False,
          save_everystep=None, exact_pow=False, cap=None, trace_cap=20000, solver="tsit5"):
    u0 = _f32(u0)
    D = fld.D; B = u0.size // D
    saveat = np.ascontiguousarray(saveat, dtype=np.float32)
    if save_everystep is None:
        save_everystep = saveat.size == 0
    if cap is None:
        cap = saveat.size + 2 + (min(maxiters, 4096) if save_everystep else 0)
    o = make_opts(abstol, reltol, maxiters, save_start, save_everystep, exact_pow, solver)
    us = np.empty((cap,) + u0.shape, dtype=np.float32)
    ts = np.empty(cap, dtype=np.float32)
    st = Stats()
    tr = (TraceRow * trace_cap)()
    rc = lib().lro_solve(C.byref(fld.field), _fp(u0), B, float(t0), float(t1), C.byref(o),
                         _fp(saveat) if saveat.size else None, int(saveat.size), _fp(us), _fp(ts),
                         cap, C.byref(st), tr, trace_cap)
    nt = st.naccept + st.nreject
    trace = np.array([(tr[i].t, tr[i].dt, tr[i].eest, tr[i].accepted) for i in range(min(nt, trace_cap))],
                     dtype=[("t", "f4"), ("dt", "f4"), ("eest", "f4"), ("accepted", "i4")])
    return dict(retcode=rc, u=us[:st.nsaved], t=ts[:st.nsaved], stats=st.asdict(), trace=trace)


def node_forward(fld, x, t0, t2, abstol, reltol, mode="unbiased", reg_type="error_estimate",
                 t1_or_rand=0.5, maxiters=1000, save_start=False, exact_pow=False, solver="tsit5"):
    x = _f32(x)
    B = x.size // fld.D
    o = make_opts(abstol, reltol, maxiters, save_start, False, exact_pow, solver)
    u_end = np.empty_like(x)
    reg = C.c_float(); nfe = C.c_int(); st = Stats(); t1u = C.c_float()
    rc = lib().lro_node_forward(C.byref(fld.field), _fp(x), B, float(t0), float(t2), C.byref(o),
                                MODE[mode], REG[reg_type], float(t1_or_rand), _fp(u_end),
                                C.byref(reg), C.byref(nfe), C.byref(st), C.byref(t1u))
    return dict(retcode=rc, u_end=u_end, reg_val=np.float32(reg.value), nfe=nfe.value,
                stats=st.asdict(), t1=np.float32(t1u.value))


def euler_heun_step(drift, diffusion, uprev, dW, t, dt, abstol, reltol, delta):
    uprev = _f32(uprev); dW = _f32(dW)
    B = uprev.size // drift.D
    u = np.empty_like(uprev)
    ee = C.c_float(); rv = C.c_float()
    lib().lro_euler_heun_step(C.byref(drift.field), C.byref(diffusion.field), _fp(uprev), _fp(dW),
                              float(t), float(dt), float(abstol), float(reltol), float(delta), B,
                              _fp(u), C.byref(ee), C.byref(rv))
    return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))


# ---------------------------------------------------------------------------------------------
# NeuralDSDE layer (src/layers/neural_sde.jl:50-123): the oracle counterpart of lrnde_sde_node_forward_record.  The step is
# the C oracle's lro_euler_heun_step; around it, in float32 numpy, the loop StochasticDiffEq runs (UPSTREAM-RECALL: that
# package is neither vendored nor readable here — "parity unpinned" for everything in this block but the step): automatic
# initial dt (sde_determine_initdt: the ODE heuristic with the diffusion entering as +-3 g), the PI controller on EEst with
# steps of whole grid intervals of the caller's Brownian path, saveat values by the linear interpolant, the layer's saveat /
# t1 rules (src/layers/neural_ode.jl:102-116, neural_sde.jl:88-123) and the local step at (sol(t1), t1).
# ---------------------------------------------------------------------------------------------
def sde_init_dt(drift, diffusion, u, t, tend, abstol, reltol, order=0.5):
    f32 = np.float32
    u = _f32(u); abstol, reltol, t, tend = f32(abstol), f32(reltol), f32(t), f32(tend)
    n = u.size
    rms = lambda r: f32(np.sqrt(np.sum((r * r).astype(np.float64)) / n))
    sk = abstol + np.abs(u) * reltol
    f0, g0 = drift.rhs(u, t), diffusion.rhs(u, t)
    G0 = f32(3) * g0
    d0 = rms(u / sk)
    d1 = rms(np.maximum(np.abs(f0 + G0), np.abs(f0 - G0)) / sk)
    dtmax = f32(tend - t)
    dt0 = f32(1e-6) if (float(d0) < 1e-5 or float(d1) < 1e-5) else f32(f32(d0 / d1) / f32(100))
    dt0 = min(dt0, dtmax)
    u1 = (u + dt0 * f0).astype(f32)
    f1, g1 = drift.rhs(u1, f32(t + dt0)), diffusion.rhs(u1, f32(t + dt0))
    G1 = f32(3) * g1
    dg = np.maximum(np.abs(G0 - G1), np.abs(G0 + G1))
    df = f1 - f0
    d2 = f32(rms(np.maximum(np.abs(df + dg), np.abs(df - dg)) / sk) / dt0)
    md = max(d1, d2)
    if float(md) <= 1e-15:
        dt1 = max(f32(1e-6), f32(dt0 * f32(1e-3)))
    else:
        e = f32(f32(-(f32(2) + f32(np.log10(float(md))))) / f32(f32(order) + f32(0.5)))
        dt1 = f32(10.0 ** float(e))
    return min(f32(f32(100) * dt0), dt1, dtmax)


def sde_node_forward(drift, diffusion, x, W, t0, t2, abstol, reltol, mode="unbiased", t1_or_rand=0.5, z_local=None, saveat=(),
                     save_start=-1, delta=1.0 / 6.0, dt0=0.0, gamma=0.9, qmin=0.2, qmax=1.125, beta1=7.0 / 50.0, beta2=2.0 / 25.0,
                     maxiters=10000):
    """dict(u (nseries,B,D), t, reg_val, nfe_drift, nfe_diffusion, naccept, nreject, steps [(i, m)], t1, dt_local, u1, dW_local)"""
    f32 = np.float32
    x = _f32(x); W = _f32(W)
    nfine = W.shape[0] - 1
    t0, t2 = f32(t0), f32(t2)
    h = f32(f32(t2 - t0) / f32(nfine))
    fp = lambda a, b: f32(lib().lro_fastpow(float(a), float(b)))
    gamma, qmin, qmax, beta1, beta2 = f32(gamma), f32(qmin), f32(qmax), f32(beta1), f32(beta2)
    nff = ngg = 0
    d0 = f32(dt0)
    if not d0 > 0:
        d0 = sde_init_dt(drift, diffusion, x, t0, t2, abstol, reltol); nff += 2; ngg += 2
    i, m, qold, u, dtc = 0, max(int(f32(d0 / h)), 1), f32(1e-4), x, f32(d0)
    steps, states, nacc, nrej, iters = [], [], 0, 0, 0
    while i < nfine:
        m = min(m, nfine - i)
        iters += 1
        assert iters <= maxiters
        t, dt = f32(t0 + f32(i) * h), f32(f32(m) * h)
        r = euler_heun_step(drift, diffusion, u, (W[i + m] - W[i]).astype(f32), t, dt, abstol, reltol, delta)
        ee = r["eest"]
        q = f32(f32(1) / qmax) if ee == 0 else max(f32(f32(1) / qmax), min(f32(f32(1) / qmin), f32(f32(fp(ee, beta1) / fp(qold, beta2)) / gamma)))
        dtc = f32((max(dtc, dt) if ee <= 1 else dt) / q)   # the proposal stays a real number; the step is its floor on the grid
        mnew = max(int(f32(dtc / h)), 1)
        if ee <= 1:
            nacc += 1; steps.append((i, m)); states.append(r["u"])
            qold, i, u, m = max(ee, f32(1e-4)), i + m, r["u"], mnew
        else:
            nrej += 1
            assert m > 1, "DtLessThanMin: the path's grid cannot be refined further"
            m = mnew if mnew < m else m - 1
    nff += 3 * (nacc + nrej); ngg += 3 * (nacc + nrej)
    K = len(steps)
    tk = lambda k: f32(t0 + f32(steps[k][0]) * h)
    tk1 = lambda k: t2 if steps[k][0] + steps[k][1] >= nfine else f32(t0 + f32(steps[k][0] + steps[k][1]) * h)

    def entry(ts):
        ts = f32(ts)
        if not ts > t0:
            return (ts, -1, f32(0))
        k = 0
        while k < K - 1 and tk1(k) < ts:
            k += 1
        th = f32(1) if ts >= tk1(k) else f32(f32(ts - tk(k)) / f32(f32(steps[k][1]) * h))
        return (ts, k, th)

    def value(e):
        ts, k, th = e
        if k < 0:
            return x
        if th == 1:
            return states[k]
        a = x if k == 0 else states[k - 1]
        return (f32(f32(1) - th) * a + th * states[k]).astype(f32)

    sv_user = [f32(v) for v in saveat]
    needs_corr = everystep = False
    t1 = t2
    if mode == "unbiased":
        t1 = f32(t1_or_rand)
        if sv_user:
            sv = sorted(sv_user + [t1]); needs_corr = True
        else:
            sv = [t1, t2]
    elif sv_user:
        sv = sv_user
    elif mode == "biased":
        sv, everystep = [], True
    else:
        sv = [t2]
    with_start = save_start > 0 if save_start >= 0 else (everystep or (len(sv) > 0 and sv[0] == t0))
    sol = [(t0, -1, f32(0))] if with_start else []
    if everystep:
        sol += [(tk1(k), k, f32(1)) for k in range(K)]
    else:
        sol += [entry(ts) for ts in sv if not (ts == t0 and with_start)]
    e1 = None
    if mode == "biased":
        mm = len(sol) - 1
        assert mm >= 1, ":biased needs at least two saved times"
        idx = min(max(int(f32(t1_or_rand) * f32(mm)), 0), mm - 1)
        e1 = sol[idx]; t1 = e1[0]
    elif mode == "unbiased":
        e1 = entry(t1)
    reg, dtl, u1, dwl = f32(0), f32(0), None, None
    if mode != "none":
        assert t1 < t2, "t1 must lie before the end of tspan"
        u1 = value(e1)
        dtl = f32(dt0)
        if not dtl > 0:
            dtl = sde_init_dt(drift, diffusion, u1, t1, t2, abstol, reltol); nff += 2; ngg += 2
        dtl = min(dtl, f32(t2 - t1))
        dwl = (f32(np.sqrt(dtl)) * _f32(z_local)).astype(f32)
        reg = euler_heun_step(drift, diffusion, u1, dwl, t1, dtl, abstol, reltol, delta)["reg_val"]
        nff += 3; ngg += 3
    series = [e for e in sol if not (needs_corr and e[0] == t1)]
    return dict(u=np.stack([value(e) for e in series]), t=np.array([e[0] for e in series], f32), reg_val=reg, nfe_drift=nff,
                nfe_diffusion=ngg, naccept=nacc, nreject=nrej, steps=steps, t1=t1, dt_local=dtl, u1=u1, dW_local=dwl, series=series,
                dt0=d0)


def glorot_mlp_params(D, H, time_dep=True, seed=0):
    """Lux Dense init: W ~ glorot_uniform = (rand-0.5)*sqrt(24/(in+out)), b = 0
    (SURVEY.md §3.5); numpy stream (the Julia RNG streams cannot be reproduced here)."""
    rng = np.random.default_rng(seed)
    td = int(time_dep)
    def glorot(out, inn):
        return ((rng.random((inn, out), dtype=np.float32) - np.float32(0.5)) *
                np.float32(np.sqrt(24.0 / (inn + out)))).astype(np.float32)  # (in,out) C-order == column-major out x in
    W1 = glorot(H, D + td); b1 = np.zeros(H, np.float32)
    W2 = glorot(D, H + td); b2 = np.zeros(D, np.float32)
    return np.concatenate([W1.ravel(), b1, W2.ravel(), b2]).astype(np.float32)


def mlp_vjp(fld, y, t, lam, want_gp=True):
    y = _f32(y); lam = _f32(lam)
    B = y.size // fld.D
    dy = np.empty_like(y)
    gp = np.zeros(fld.params.size, np.float32) if want_gp else None
    lib().lro_mlp_vjp(C.byref(fld.m), _fp(y), float(t), _fp(lam), B, _fp(dy), _fp(gp))
    return dy, gp


def step_reg_grad(fld, uprev, k1, t, dt, abstol, reltol, reg_type="error_estimate"):
    uprev = _f32(uprev); k1 = _f32(k1)
    B = uprev.size // fld.D
    gp = np.zeros(fld.params.size, np.float32)
    rv = C.c_float()
    fn = lib().lro_conv_step_reg_grad if isinstance(fld, ConvField) else lib().lro_tsit5_step_reg_grad
    rc = fn(C.byref(fld.m), _fp(uprev), _fp(k1), float(t), float(dt), float(abstol),
            float(reltol), B, REG[reg_type], _fp(gp), C.byref(rv))
    assert rc == 0
    return gp, np.float32(rv.value)


def node_backward(fld, x, t0, t2, abstol, reltol, du_end, mode="unbiased", reg_type="error_estimate",
                  t1_or_rand=0.5, w_reg=0.0, maxiters=10000, save_start=False, trace=False, solver="tsit5"):
    x = _f32(x); du_end = _f32(du_end)
    B = x.size // fld.D
    o = make_opts(abstol, reltol, maxiters, save_start, False, False, solver)
    dx = np.empty_like(x)
    dp = np.zeros(fld.params.size, np.float32)
    sf, sb = Stats(), Stats()
    if trace and not isinstance(fld, ConvField):
        cap = maxiters + 2
        rows = (TraceRow * cap)()
        rc = lib().lro_node_backward_traced(C.byref(fld.m), _fp(x), B, float(t0), float(t2), C.byref(o), MODE[mode],
                                            REG[reg_type], float(t1_or_rand), _fp(du_end), float(w_reg), _fp(dx), _fp(dp),
                                            C.byref(sf), C.byref(sb), rows, cap)
        tr = [(rows[i].t, rows[i].dt, rows[i].eest, rows[i].accepted) for i in range(min(cap, sb.iters))]
        return dict(retcode=rc, dx=dx, dp=dp, stats_fwd=sf.asdict(), stats_bwd=sb.asdict(), trace_bwd=tr)
    fn = lib().lro_conv_node_backward if isinstance(fld, ConvField) else lib().lro_node_backward
    rc = fn(C.byref(fld.m), _fp(x), B, float(t0), float(t2), C.byref(o), MODE[mode],
                                 REG[reg_type], float(t1_or_rand), _fp(du_end), float(w_reg), _fp(dx), _fp(dp),
                                 C.byref(sf), C.byref(sb))
    return dict(retcode=rc, dx=dx, dp=dp, stats_fwd=sf.asdict(), stats_bwd=sb.asdict())


def classifier_ce(u, pc, K, labels):
    """Dense(D => K) + logitcrossentropy: (loss, logits, du, dpc)."""
    u = _f32(u); pc = _f32(pc)
    B, D = u.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    logits = np.empty((B, K), np.float32); du = np.empty_like(u); dpc = np.empty_like(pc)
    loss = lib().lro_classifier_ce(_fp(u), B, D, _fp(pc), K, lab.ctypes.data_as(C.POINTER(C.c_int)), _fp(logits),
                                   _fp(du), _fp(dpc))
    return np.float32(loss), logits, du, dpc


def rkmil_step(drift, diffusion, uprev, dW, t, dt, abstol, reltol):
    uprev = _f32(uprev); dW = _f32(dW)
    B = uprev.size // drift.D
    u = np.empty_like(uprev)
    ee, rv = C.c_float(), C.c_float()
    rc = lib().lro_rkmil_step(C.byref(drift.field), C.byref(diffusion.field), _fp(uprev), _fp(dW), float(t), float(dt),
                              float(abstol), float(reltol), B, _fp(u), C.byref(ee), C.byref(rv))
    assert rc == 0
    return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))


SRI_FIELDS = ("a021 a031 a032 a041 a042 a043 a121 a131 a132 a141 a142 a143 "
              "b021 b031 b032 b041 b042 b043 b121 b131 b132 b141 b142 b143 "
              "c02 c03 c04 c11 c12 c13 c14 alpha1 alpha2 alpha3 alpha4 "
              "beta11 beta12 beta13 beta14 beta21 beta22 beta23 beta24 beta31 beta32 beta33 beta34 beta41 beta42 beta43 beta44").split()


class SriTableau(C.Structure):
    _fields_ = [(n, C.c_float) for n in SRI_FIELDS]


def sri_step(drift, diffusion, tableau, uprev, dW, dZ, t, dt, abstol, reltol, delta):
    """src/perform_step.jl:49-106 with a caller-supplied tableau (dict keyed by SRI_FIELDS)"""
    uprev = _f32(uprev); dW = _f32(dW); dZ = _f32(dZ)
    B = uprev.size // drift.D
    u = np.empty_like(uprev)
    ee, rv = C.c_float(), C.c_float()
    tab = SriTableau(*[float(tableau[k]) for k in SRI_FIELDS])
    L = lib()
    L.lro_sri_step.restype = C.c_int
    L.lro_sri_step.argtypes = [C.POINTER(Field), C.POINTER(Field), C.POINTER(SriTableau), C.POINTER(C.c_float), C.POINTER(C.c_float),
                               C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                               C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rc = L.lro_sri_step(C.byref(drift.field), C.byref(diffusion.field), C.byref(tab), _fp(uprev), _fp(dW), _fp(dZ), float(t),
                        float(dt), float(abstol), float(reltol), float(delta), B, _fp(u), C.byref(ee), C.byref(rv))
    assert rc == 0
    return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))


def cifar_stem_forward(x, ps, bn_train=True, bn_state=None, eps=1e-5, return_state=False):
    """AugmenterLayer(Conv 3=>5) + BatchNorm(8): x (B,3,H,W) -> u0 (B,8,H,W) [, running statistics after the call]"""
    x = _f32(x); ps = _f32(ps)
    B, _, H, W = x.shape
    u0 = np.empty((B, 8, H, W), np.float32)
    st = None if bn_state is None else _f32(bn_state)
    st_out = np.empty(16, np.float32) if return_state else None
    lib().lro_cifar_stem_forward(_fp(x), B, H, W, _fp(ps), int(bn_train), _fp(st), float(eps), _fp(u0), _fp(st_out))
    return (u0, st_out) if return_state else u0


def cifar_stem_backward(x, ps, du0, bn_train=True, bn_state=None, eps=1e-5):
    x = _f32(x); ps = _f32(ps); du0 = _f32(du0)
    B, _, H, W = x.shape
    dps = np.zeros(156, np.float32)
    st = None if bn_state is None else _f32(bn_state)
    lib().lro_cifar_stem_backward(_fp(x), B, H, W, _fp(ps), int(bn_train), _fp(st), float(eps), _fp(du0), _fp(dps))
    return dps


def cifar_head_ce(u, ph, K, labels):
    """Conv(8=>1, gelu) + flatten + Dense(H*W=>K) + logitcrossentropy: (loss, logits, du, dph)"""
    u = _f32(u); ph = _f32(ph)
    B, _, H, W = u.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    logits = np.empty((B, K), np.float32); du = np.empty_like(u); dph = np.zeros_like(ph)
    loss = lib().lro_cifar_head_ce(_fp(u), B, H, W, _fp(ph), K, lab.ctypes.data_as(C.POINTER(C.c_int)), _fp(logits), _fp(du), _fp(dph))
    return np.float32(loss), logits, du, dph
