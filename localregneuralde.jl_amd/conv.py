"""Host mirror of the conv vector field (experiments/src/construct.jl:213-218) over the
lrnde_conv_* C ABI (include/lrnde.h): layer specs `Conv`, `BatchNorm` and `ConvHandle`, which has
the same methods as `layers.Handle` so `NeuralODE` drives either field."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class Conv:
    """Lux.Conv((3,3), in => out; pad=(1,1), use_bias=false) as a shape spec."""

    def __init__(self, kernel, in_chs, out_chs, activation="identity", pad=(1, 1), use_bias=False):
        if tuple(kernel) != (3, 3) or tuple(pad) != (1, 1) or use_bias or activation != "identity":
            raise NotImplementedError("the reference's node_core uses Conv((3,3), pad=(1,1), use_bias=false) only")
        self.in_chs, self.out_chs = int(in_chs), int(out_chs)


class BatchNorm:
    """Lux.BatchNorm(chs, activation) (affine, track_stats; epsilon = 1f-5)."""

    def __init__(self, chs, activation="identity", epsilon=1e-5):
        if activation not in L.ACT:
            raise ValueError(f"unsupported activation {activation!r}")
        self.chs, self.activation, self.epsilon = int(chs), activation, float(epsilon)


def conv_topology(model):
    """(C, Hc, act, eps) if `model` is TDChain(Chain(Conv(C+1=>Hc), BN(Hc,act)), Chain(Conv(Hc+1=>Hc), BN(Hc,act)),
    Conv(Hc+1=>C)), else None."""
    from .layers import Chain, TDChain
    if not isinstance(model, TDChain) or len(model.layers) != 3:
        return None
    l1, l2, l3 = model.layers
    if not (isinstance(l1, Chain) and isinstance(l2, Chain) and isinstance(l3, Conv)):
        return None
    for blk in (l1, l2):
        if len(blk.layers) != 2 or not isinstance(blk.layers[0], Conv) or not isinstance(blk.layers[1], BatchNorm):
            return None
    c1, b1, c2, b2 = l1.layers[0], l1.layers[1], l2.layers[0], l2.layers[1]
    Cst, Hc = c1.in_chs - 1, c1.out_chs
    ok = (b1.chs == Hc and c2.in_chs == Hc + 1 and c2.out_chs == Hc and b2.chs == Hc and l3.in_chs == Hc + 1 and
          l3.out_chs == Cst and b1.activation == b2.activation and b1.epsilon == b2.epsilon)
    if not ok:
        raise ValueError("conv shapes do not chain: need Conv(C+1=>Hc),BN(Hc) / Conv(Hc+1=>Hc),BN(Hc) / Conv(Hc+1=>C)")
    return Cst, Hc, b1.activation, b1.epsilon


def glorot_conv_params(Cst, Hc, seed=0):
    """Lux default init for the node_core (glorot_uniform conv weights, BatchNorm scale 1 / bias 0), flat
    ComponentArray order, from a numpy stream."""
    rng = np.random.default_rng(seed)

    def glorot(cin, cout):
        return ((rng.random(9 * cin * cout, dtype=np.float32) - np.float32(0.5)) *
                np.float32(np.sqrt(24.0 / (9 * cin + 9 * cout)))).astype(np.float32)

    one, zero = np.ones(Hc, np.float32), np.zeros(Hc, np.float32)
    return np.concatenate([glorot(Cst + 1, Hc), one, zero, glorot(Hc + 1, Hc), one, zero, glorot(Hc + 1, Cst)])


def _ptr(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")
    return C.c_void_p(t.data_ptr())


class ConvHandle:
    """One lrnde_conv context.  States are (B, C, H, W) contiguous float32 CUDA tensors — the memory order of
    the reference's Julia (W, H, C, B) arrays."""

    def __init__(self, width, height, channels=8, hidden=64, act="gelu", bn_train=True, compute_dtype="f32",
                 bn_eps=1e-5, device=None, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("liblrnde needs a GPU (gfx950); there is no CPU fallback")
        self.desc = L.ConvDesc(int(width), int(height), int(channels), int(hidden), L.ACT[act], int(bool(bn_train)),
                               L.DTYPE[compute_dtype], float(bn_eps))
        self.D = int(width) * int(height) * int(channels)
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._stream = torch.cuda.current_stream(self.device) if stream is None else stream
        self._ctx = C.c_void_p()
        rc = L.lib.lrnde_conv_create(C.byref(self._ctx), C.byref(self.desc), self.device,
                                     C.c_void_p(self._stream.cuda_stream))
        if rc != 0:
            raise L.LrndeError(rc, "lrnde_conv_create failed (shape or dtype not supported by the kernels)")
        self._keep = {}  # name -> the tensor a pending async copy reads (one slot per kind: overwritten, never appended)

    def close(self):
        if getattr(self, "_ctx", None):
            L.lib.lrnde_conv_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = L.lib.lrnde_conv_last_error(self._ctx)
            raise L.LrndeError(rc, msg.decode() if msg else "")

    def _B(self, u):
        if u.numel() % self.D:
            raise ValueError(f"state has {u.numel()} elements, not a multiple of W*H*C = {self.D}")
        return u.numel() // self.D

    def param_count(self):
        return int(L.lib.lrnde_conv_param_count(C.byref(self.desc)))

    def set_params(self, ps):
        ps = ps if isinstance(ps, torch.Tensor) else torch.as_tensor(np.asarray(ps, dtype=np.float32))
        ps = ps.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous().reshape(-1)
        self._keep["params"] = ps
        self._chk(L.lib.lrnde_conv_set_params(self._ctx, C.c_void_p(ps.data_ptr()), ps.numel()))

    def set_bn_state(self, mean_var):
        mv = mean_var if isinstance(mean_var, torch.Tensor) else torch.as_tensor(np.asarray(mean_var, dtype=np.float32))
        mv = mv.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous().reshape(-1)
        self._keep["bn_state"] = mv
        self._chk(L.lib.lrnde_conv_set_bn_state(self._ctx, C.c_void_p(mv.data_ptr()), mv.numel()))

    def set_bn_mode(self, train):
        """Lux.trainmode / Lux.testmode of the BatchNorm layers"""
        self._chk(L.lib.lrnde_conv_set_bn_mode(self._ctx, int(bool(train))))

    def get_bn_state(self):
        """running [mean1 var1 mean2 var2] (st.model of the layer), advanced by every training-mode f-eval"""
        out = torch.empty(4 * self.desc.hidden, dtype=torch.float32, device=f"cuda:{self.device}")
        self._chk(L.lib.lrnde_conv_get_bn_state(self._ctx, C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def rhs(self, u, t):
        du = torch.empty_like(u)
        self._chk(L.lib.lrnde_conv_rhs(self._ctx, _ptr(u, "u"), float(t), self._B(u), _ptr(du, "du")))
        return du

    def init_dt(self, u0, t0, tend, abstol, reltol):
        k1 = torch.empty_like(u0)
        dt = C.c_float()
        self._chk(L.lib.lrnde_conv_init_dt(self._ctx, _ptr(u0, "u0"), self._B(u0), float(t0), float(tend), float(abstol),
                                           float(reltol), _ptr(k1, "k1"), C.byref(dt)))
        return np.float32(dt.value), k1

    def perform_step(self, uprev, k1, t, dt, abstol, reltol):
        u, k7 = torch.empty_like(uprev), torch.empty_like(uprev)
        ee, re, rs = C.c_float(), C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_conv_perform_step(self._ctx, _ptr(uprev, "uprev"), _ptr(k1, "k1"), self._B(uprev), float(t),
                                                float(dt), float(abstol), float(reltol), _ptr(u, "u"), _ptr(k7, "k7"),
                                                C.byref(ee), C.byref(re), C.byref(rs)))
        return dict(u=u, k7=k7, eest=np.float32(ee.value), reg_error=np.float32(re.value),
                    reg_stiff=np.float32(rs.value))

    def solve(self, u0, t0, t1, abstol, reltol, saveat=(), maxiters=1000, save_start=False, save_everystep=None,
              exact_pow=False, cap=None, trace=False, raise_on_retcode=True):
        sv = np.ascontiguousarray(saveat, dtype=np.float32)
        if save_everystep is None:
            save_everystep = sv.size == 0
        if cap is None:
            cap = int(sv.size) + 2 + (min(int(maxiters), 512) if save_everystep else 0)
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), int(save_everystep), int(exact_pow))
        us = torch.empty((cap,) + tuple(u0.shape), dtype=torch.float32, device=u0.device)
        ts = np.empty(cap, dtype=np.float32)
        st = L.Stats()
        ntr = int(maxiters) + 8 if trace else 0
        tr = (L.TraceRow * max(ntr, 1))()
        rc = L.lib.lrnde_conv_solve(self._ctx, _ptr(u0, "u0"), self._B(u0), float(t0), float(t1), C.byref(o),
                                    sv.ctypes.data_as(C.POINTER(C.c_float)) if sv.size else None, int(sv.size),
                                    C.c_void_p(us.data_ptr()), ts.ctypes.data_as(C.POINTER(C.c_float)), cap,
                                    C.byref(st), tr if trace else None, ntr)
        if rc != 0 and (raise_on_retcode or rc >= 4):
            self._chk(rc)
        out = dict(retcode=rc, u=us[:st.nsaved], t=ts[:st.nsaved].copy(), stats=st.asdict())
        if trace:
            nt = min(st.naccept + st.nreject, ntr)
            out["trace"] = np.array([(tr[i].t, tr[i].dt, tr[i].eest, tr[i].accepted) for i in range(nt)],
                                    dtype=[("t", "f4"), ("dt", "f4"), ("eest", "f4"), ("accepted", "i4")])
        return out

    def node_forward(self, x, t0, t2, abstol, reltol, mode="unbiased", reg_type="error_estimate", t1_or_rand=0.5,
                     maxiters=1000, save_start=False, exact_pow=False):
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        u_end = torch.empty_like(x)
        reg, nfe, st, t1u = C.c_float(), C.c_int32(), L.Stats(), C.c_float()
        self._chk(L.lib.lrnde_conv_node_forward(self._ctx, _ptr(x, "x"), self._B(x), float(t0), float(t2), C.byref(o),
                                                L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand),
                                                _ptr(u_end, "u_end"), C.byref(reg), C.byref(nfe), C.byref(st),
                                                C.byref(t1u)))
        return dict(u_end=u_end, reg_val=np.float32(reg.value), nfe=int(nfe.value), stats=st.asdict(),
                    t1=np.float32(t1u.value))

    def vjp(self, y, t, lam, want_gp=True):
        """(J^T lam, (df/dp)^T lam) of the conv field at (y, t) — the adjoint RHS building block."""
        dy = torch.empty_like(y)
        gp = torch.zeros(self.param_count(), dtype=torch.float32, device=y.device) if want_gp else None
        self._chk(L.lib.lrnde_conv_vjp(self._ctx, _ptr(y, "y"), float(t), _ptr(lam, "lam"), self._B(y), _ptr(dy, "dy"),
                                       C.c_void_p(gp.data_ptr()) if want_gp else None))
        return dy, gp

    def step_reg_grad(self, uprev, k1, t, dt, abstol, reltol, reg_type="error_estimate"):
        """d reg_val / d ps of one local Tsit5 step (k1, dt, uprev constant)."""
        gp = torch.empty(self.param_count(), dtype=torch.float32, device=uprev.device)
        rv = C.c_float()
        self._chk(L.lib.lrnde_conv_step_reg_grad(self._ctx, _ptr(uprev, "uprev"), _ptr(k1, "k1"), self._B(uprev), float(t),
                                                 float(dt), float(abstol), float(reltol), L.REG_TYPE[reg_type],
                                                 C.c_void_p(gp.data_ptr()), C.byref(rv)))
        return gp, np.float32(rv.value)

    def node_backward(self, x, t0, t2, abstol, reltol, du_end, mode="unbiased", reg_type="error_estimate", t1_or_rand=0.5,
                      w_reg=0.0, maxiters=1000, save_start=False, exact_pow=False):
        """Pullback of the layer for loss = <du_end, sol.u[end]> + w_reg*reg_val -> (dx, dp)."""
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        dx = torch.empty_like(x)
        dp = torch.empty(self.param_count(), dtype=torch.float32, device=x.device)
        sf, sb = L.Stats(), L.Stats()
        self._chk(L.lib.lrnde_conv_node_backward(self._ctx, _ptr(x, "x"), self._B(x), float(t0), float(t2), C.byref(o),
                                                 L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand), _ptr(du_end, "du_end"),
                                                 float(w_reg), _ptr(dx, "dx"), C.c_void_p(dp.data_ptr()), C.byref(sf),
                                                 C.byref(sb)))
        return dict(dx=dx, dp=dp, stats_fwd=sf.asdict(), stats_bwd=sb.asdict())

    def node_forward_record(self, x, t0, t2, abstol, reltol, mode="unbiased", reg_type="error_estimate", t1_or_rand=0.5,
                            maxiters=1000, save_start=False, exact_pow=False):
        """node_forward that keeps the record node_backward_recorded pulls back through (one forward per training step)"""
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        u_end = torch.empty_like(x)
        reg, nfe, st, t1u = C.c_float(), C.c_int32(), L.Stats(), C.c_float()
        self._chk(L.lib.lrnde_conv_node_forward_record(self._ctx, _ptr(x, "x"), self._B(x), float(t0), float(t2), C.byref(o),
                                                       L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand),
                                                       _ptr(u_end, "u_end"), C.byref(reg), C.byref(nfe), C.byref(st),
                                                       C.byref(t1u)))
        return dict(u_end=u_end, reg_val=np.float32(reg.value), nfe=int(nfe.value), stats=st.asdict(),
                    t1=np.float32(t1u.value))

    def node_backward_recorded(self, B, du_end, w_reg=0.0):
        """(dx, dp) of <du_end, sol.u[end]> + w_reg*reg_val from the record of the last node_forward_record"""
        dx = torch.empty_like(du_end)
        dp = torch.empty(self.param_count(), dtype=torch.float32, device=du_end.device)
        sb = L.Stats()
        self._chk(L.lib.lrnde_conv_node_backward_recorded(self._ctx, int(B), _ptr(du_end, "du_end"), float(w_reg), _ptr(dx, "dx"),
                                                          C.c_void_p(dp.data_ptr()), C.byref(sb)))
        return dict(dx=dx, dp=dp, stats_bwd=sb.asdict())

    # ---- layers around the CIFAR10 NeuralODE (experiments/src/construct.jl:224-227) ----
    def cifar_stem_forward(self, x, ps, bn_state=None, return_state=False):
        """AugmenterLayer(Conv 3=>5) + BatchNorm(8): x (B,3,H,W) -> u0 (B,8,H,W); with return_state also the layer's
        running statistics [mean 8; var 8] after this call (advanced in training mode, as Lux returns them in `st`)"""
        B = x.shape[0]
        u0 = torch.empty((B, 8, self.desc.height, self.desc.width), dtype=torch.float32, device=x.device)
        st_out = torch.empty(16, dtype=torch.float32, device=x.device) if return_state else None
        self._chk(L.lib.lrnde_cifar_stem_forward(self._ctx, _ptr(x, "x"), B, _ptr(ps, "ps"),
                                                 _ptr(bn_state, "bn_state") if bn_state is not None else None, _ptr(u0, "u0"),
                                                 _ptr(st_out, "bn_state_out") if return_state else None))
        return (u0, st_out) if return_state else u0

    def cifar_stem_backward(self, x, ps, du0, bn_state=None):
        dps = torch.empty(156, dtype=torch.float32, device=x.device)
        self._chk(L.lib.lrnde_cifar_stem_backward(self._ctx, _ptr(x, "x"), x.shape[0], _ptr(ps, "ps"),
                                                  _ptr(bn_state, "bn_state") if bn_state is not None else None, _ptr(du0, "du0"),
                                                  _ptr(dps, "dps")))
        return dps

    def cifar_head_ce(self, u, ph, K, labels, want_grads=True):
        """Conv(8=>1, gelu) + flatten + Dense(H*W=>K) + logitcrossentropy: dict(loss, logits, du, dph)"""
        B = u.shape[0]
        if not (labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == B):
            raise ValueError("labels must be a CUDA int32 tensor of length B")
        logits = torch.empty((B, K), dtype=torch.float32, device=u.device)
        du = torch.empty_like(u) if want_grads else None
        dph = torch.empty_like(ph) if want_grads else None
        loss = C.c_float()
        self._chk(L.lib.lrnde_cifar_head_ce(self._ctx, _ptr(u, "u"), B, _ptr(ph, "ph"), int(K), C.c_void_p(labels.data_ptr()),
                                            C.byref(loss), C.c_void_p(logits.data_ptr()),
                                            C.c_void_p(du.data_ptr()) if want_grads else None,
                                            C.c_void_p(dph.data_ptr()) if want_grads else None))
        return dict(loss=np.float32(loss.value), logits=logits, du=du, dph=dph)

    def bench_rhs(self, u, t, reps=20):
        """average microseconds of one f-eval (HIP events on the handle's stream)."""
        us = C.c_float()
        self._chk(L.lib.lrnde_conv_bench_rhs(self._ctx, _ptr(u, "u"), float(t), self._B(u), int(reps), C.byref(us)))
        return float(us.value)
