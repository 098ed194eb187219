"""NeuralODE layer mirror (src/layers/neural_ode.jl) over the liblrnde C ABI.

States are torch.cuda float32 tensors of shape (B, D), contiguous: the same
memory as the reference's column-major D x B Julia arrays (batch last).
"""
import copy
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib as L

_VALID_REGULARIZE = ("none", "unbiased", "biased")
_VALID_REG_TYPES = ("error_estimate", "stiffness_estimate")


def _sym(s):
    return s[1:] if isinstance(s, str) and s.startswith(":") else s


def _check_valid_regularize(regularize, valid_modes=_VALID_REGULARIZE):
    """src/utils.jl:53-58 — ArgumentError -> ValueError with the same text."""
    if regularize not in valid_modes:
        names = ", ".join(":" + m for m in valid_modes)
        raise ValueError(f"regularize must be one of ({names})")


# ----- model description (shape source: experiments/src/construct.jl:180-189) -----
class Dense:
    """Lux.Dense(in => out, activation) as a shape/activation spec."""

    def __init__(self, in_dims, out_dims, activation="identity"):
        if activation not in L.ACT:
            raise ValueError(f"unsupported activation {activation!r} (have {sorted(L.ACT)})")
        self.in_dims, self.out_dims, self.activation = int(in_dims), int(out_dims), activation


class Chain:
    def __init__(self, *layers):
        self.layers = list(layers)


class TDChain:
    """src/layers/common.jl:2-45 — `t` is concatenated to the input of EVERY sub-layer."""

    def __init__(self, chain):
        self.layers = list(chain.layers if isinstance(chain, Chain) else chain)


def _mlp_desc(model):
    td = isinstance(model, TDChain)
    if not isinstance(model, (Chain, TDChain)) or len(model.layers) != 2 or \
            not all(isinstance(l, Dense) for l in model.layers):
        raise NotImplementedError("liblrnde implements the 2-layer Dense vector field "
                                  "(TDChain/Chain(Dense, Dense)) of experiments/src/construct.jl:180-189")
    l1, l2 = model.layers
    D, H = l1.in_dims - int(td), l1.out_dims
    if l2.in_dims != H + int(td) or l2.out_dims != D:
        raise ValueError("Dense shapes do not chain: need (D+td => H), (H+td => D)")
    if l2.activation != "identity":
        raise NotImplementedError("activation on the output Dense is not part of the reference fields")
    return L.ModelDesc(D, H, int(td), L.ACT[l1.activation])


def flatten_params(W1, b1, W2, b2):
    """(out, in) torch/numpy weights -> the flat Lux/ComponentArray vector (column-major vec)."""
    parts = [torch.as_tensor(W1).t().reshape(-1), torch.as_tensor(b1).reshape(-1),
             torch.as_tensor(W2).t().reshape(-1), torch.as_tensor(b2).reshape(-1)]
    return torch.cat([p.to(torch.float32) for p in parts]).contiguous()


def glorot_params(model, seed=0):
    """Lux default init (glorot_uniform weights, zero bias) from a numpy stream — the Julia RNG
    streams cannot be reproduced, so parity tests pass the same vector to both sides."""
    d = _mlp_desc(model)
    rng = np.random.default_rng(seed)
    td = d.time_dep

    def glorot(out, inn):
        return ((rng.random((inn, out), dtype=np.float32) - np.float32(0.5)) *
                np.float32(np.sqrt(24.0 / (inn + out)))).astype(np.float32)

    W1 = glorot(d.hidden_dim, d.state_dim + td)
    W2 = glorot(d.state_dim, d.hidden_dim + td)
    return np.concatenate([W1.ravel(), np.zeros(d.hidden_dim, np.float32), W2.ravel(),
                           np.zeros(d.state_dim, np.float32)])


def _dev_ptr(t, name, shape_tail=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")
    if shape_tail is not None and (t.dim() < 1 or t.shape[-1] != shape_tail):
        raise ValueError(f"{name} must have trailing dimension {shape_tail} (got {tuple(t.shape)})")
    return C.c_void_p(t.data_ptr())


# ----- thin object over lrnde_ctx -----
SOLVERS = {"tsit5": 0, "vcab3": 1, "vcabm3": 2}


def _solver_name(solver):
    """"Tsit5" / "tsit5" / "Tsit5()" ... -> the key of SOLVERS (ArgumentError of `_ode_solver` otherwise)."""
    name = str(solver).strip().rstrip("()").lower()
    if name not in SOLVERS:
        raise ValueError("unknown SolverConfig.")   # experiments/src/construct.jl:163
    return name


class Handle:
    def __init__(self, desc, device=None, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("liblrnde needs a GPU (gfx950); there is no CPU fallback")
        self.desc = desc
        self.D = desc.state_dim
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._stream = torch.cuda.current_stream(self.device) if stream is None else stream
        self._ctx = C.c_void_p()
        rc = L.lib.lrnde_create(C.byref(self._ctx), C.byref(desc), self.device,
                                C.c_void_p(self._stream.cuda_stream))
        if rc != 0:
            raise L.LrndeError(rc, "lrnde_create failed")
        self._params = None

    def close(self):
        if getattr(self, "_ctx", None):
            L.lib.lrnde_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        L.check(self._ctx, rc)

    def set_params(self, ps):
        ps = ps if isinstance(ps, torch.Tensor) else torch.as_tensor(np.asarray(ps, dtype=np.float32))
        ps = ps.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous().reshape(-1)
        self._params = ps  # keep alive until packed
        self._chk(L.lib.lrnde_set_params(self._ctx, C.c_void_p(ps.data_ptr()), ps.numel()))

    def set_solver(self, solver):
        """n.solver of the layer's global solve (experiments/src/construct.jl:154-164): "tsit5" | "vcab3" | "vcabm3"."""
        self._chk(L.lib.lrnde_set_solver(self._ctx, SOLVERS[_solver_name(solver)]))

    def rhs(self, u, t):
        B = u.numel() // self.D
        du = torch.empty_like(u)
        self._chk(L.lib.lrnde_rhs(self._ctx, _dev_ptr(u, "u", self.D), float(t), B, _dev_ptr(du, "du")))
        return du

    def init_dt(self, u0, t0, tend, abstol, reltol):
        B = u0.numel() // self.D
        k1 = torch.empty_like(u0)
        dt = C.c_float()
        self._chk(L.lib.lrnde_init_dt(self._ctx, _dev_ptr(u0, "u0", self.D), B, float(t0), float(tend),
                                      float(abstol), float(reltol), _dev_ptr(k1, "k1"), C.byref(dt)))
        return np.float32(dt.value), k1

    def perform_step(self, uprev, k1, t, dt, abstol, reltol):
        B = uprev.numel() // self.D
        u = torch.empty_like(uprev)
        k7 = torch.empty_like(uprev)
        ee, re, rs = C.c_float(), C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_perform_step(self._ctx, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(k1, "k1", self.D),
                                           B, float(t), float(dt), float(abstol), float(reltol),
                                           _dev_ptr(u, "u"), _dev_ptr(k7, "k7"), C.byref(ee), C.byref(re),
                                           C.byref(rs)))
        return dict(u=u, k7=k7, eest=np.float32(ee.value), reg_error=np.float32(re.value),
                    reg_stiff=np.float32(rs.value))

    def solve(self, u0, t0, t1, abstol, reltol, saveat=(), maxiters=1000, save_start=False,
              save_everystep=None, exact_pow=False, cap=None, trace=False, raise_on_retcode=True):
        B = u0.numel() // self.D
        sv = np.ascontiguousarray(saveat, dtype=np.float32)
        if save_everystep is None:
            save_everystep = sv.size == 0
        if cap is None:
            cap = int(sv.size) + 2 + (min(int(maxiters), 512) if save_everystep else 0)
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start),
                        int(save_everystep), int(exact_pow))
        us = torch.empty((cap,) + tuple(u0.shape), dtype=torch.float32, device=u0.device)
        ts = np.empty(cap, dtype=np.float32)
        st = L.Stats()
        ntr = int(maxiters) + 8 if trace else 0
        tr = (L.TraceRow * max(ntr, 1))()
        rc = L.lib.lrnde_solve(self._ctx, _dev_ptr(u0, "u0", self.D), B, float(t0), float(t1), C.byref(o),
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

    def node_forward(self, x, t0, t2, abstol, reltol, mode="unbiased", reg_type="error_estimate",
                     t1_or_rand=0.5, maxiters=1000, save_start=False, exact_pow=False):
        B = x.numel() // self.D
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        u_end = torch.empty_like(x)
        reg, nfe, st, t1u = C.c_float(), C.c_int32(), L.Stats(), C.c_float()
        self._chk(L.lib.lrnde_node_forward(self._ctx, _dev_ptr(x, "x", self.D), B, float(t0), float(t2),
                                           C.byref(o), L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand),
                                           _dev_ptr(u_end, "u_end"), C.byref(reg), C.byref(nfe), C.byref(st),
                                           C.byref(t1u)))
        return dict(u_end=u_end, reg_val=np.float32(reg.value), nfe=int(nfe.value), stats=st.asdict(),
                    t1=np.float32(t1u.value))

    def vjp(self, y, t, lam, want_gp=True):
        """(J^T lam, (df/dp)^T lam) of the vector field at (y, t) — the adjoint RHS building block."""
        B = y.numel() // self.D
        dy = torch.empty_like(y)
        gp = torch.zeros(int(L.lib.lrnde_param_count(C.byref(self.desc))), dtype=torch.float32, device=y.device) if want_gp else None
        self._chk(L.lib.lrnde_vjp(self._ctx, _dev_ptr(y, "y", self.D), float(t), _dev_ptr(lam, "lam", self.D), B,
                                  _dev_ptr(dy, "dy"), C.c_void_p(gp.data_ptr()) if want_gp else None))
        return dy, gp

    def step_reg_grad(self, uprev, k1, t, dt, abstol, reltol, reg_type="error_estimate"):
        """d reg_val / d ps of one local Tsit5 step (k1, dt, uprev constant)."""
        B = uprev.numel() // self.D
        gp = torch.empty(int(L.lib.lrnde_param_count(C.byref(self.desc))), dtype=torch.float32, device=uprev.device)
        rv = C.c_float()
        self._chk(L.lib.lrnde_step_reg_grad(self._ctx, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(k1, "k1", self.D), B,
                                            float(t), float(dt), float(abstol), float(reltol), L.REG_TYPE[reg_type],
                                            C.c_void_p(gp.data_ptr()), C.byref(rv)))
        return gp, np.float32(rv.value)

    def node_backward(self, x, t0, t2, abstol, reltol, du_end, mode="unbiased", reg_type="error_estimate",
                      t1_or_rand=0.5, w_reg=0.0, maxiters=1000, save_start=False, exact_pow=False):
        """Pullback of the layer for loss = <du_end, sol.u[end]> + w_reg*reg_val -> (dx, dp)."""
        B = x.numel() // self.D
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        dx = torch.empty_like(x)
        dp = torch.empty(int(L.lib.lrnde_param_count(C.byref(self.desc))), dtype=torch.float32, device=x.device)
        sf, sb = L.Stats(), L.Stats()
        self._chk(L.lib.lrnde_node_backward(self._ctx, _dev_ptr(x, "x", self.D), B, float(t0), float(t2), C.byref(o),
                                            L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand),
                                            _dev_ptr(du_end, "du_end", self.D), float(w_reg), _dev_ptr(dx, "dx"),
                                            C.c_void_p(dp.data_ptr()), C.byref(sf), C.byref(sb)))
        return dict(dx=dx, dp=dp, stats_fwd=sf.asdict(), stats_bwd=sb.asdict())

    def set_adjoint_trace(self, cap):
        """diagnostic hook (include/lrnde_hooks.h): keep one (s, dt, EEst, accepted) row per attempted step of the adjoint
        solves that follow; cap = 0 switches it off"""
        self._adj_rows = (L.TraceRow * int(cap))() if cap else None
        self._chk(L.lib.lrnde_set_adjoint_trace(self._ctx, self._adj_rows, int(cap)))

    def adjoint_trace(self):
        n = C.c_int32()
        self._chk(L.lib.lrnde_adjoint_trace_rows(self._ctx, C.byref(n)))
        r = self._adj_rows
        return [(r[i].t, r[i].dt, r[i].eest, r[i].accepted) for i in range(int(n.value))]

    def node_forward_record(self, x, t0, t2, abstol, reltol, mode="unbiased", reg_type="error_estimate",
                            t1_or_rand=0.5, maxiters=1000, save_start=False, exact_pow=False):
        """node_forward that keeps the dense record for one `node_backward_recorded`."""
        B = x.numel() // self.D
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        u_end = torch.empty_like(x)
        reg, nfe, st, t1u = C.c_float(), C.c_int32(), L.Stats(), C.c_float()
        self._chk(L.lib.lrnde_node_forward_record(self._ctx, _dev_ptr(x, "x", self.D), B, float(t0), float(t2),
                                                  C.byref(o), L.MODE[mode], L.REG_TYPE[reg_type], float(t1_or_rand),
                                                  _dev_ptr(u_end, "u_end"), C.byref(reg), C.byref(nfe), C.byref(st),
                                                  C.byref(t1u)))
        return dict(u_end=u_end, reg_val=np.float32(reg.value), nfe=int(nfe.value), stats=st.asdict(),
                    t1=np.float32(t1u.value))

    def record_generation(self):
        """which recorded forward the handle's record belongs to (lrnde_record_generation); 0 = no usable record"""
        g = C.c_uint64()
        self._chk(L.lib.lrnde_record_generation(self._ctx, C.byref(g)))
        return int(g.value)

    def node_backward_recorded(self, du_end, w_reg=0.0):
        B = du_end.numel() // self.D
        dx = torch.empty_like(du_end)
        dp = torch.empty(int(L.lib.lrnde_param_count(C.byref(self.desc))), dtype=torch.float32, device=du_end.device)
        sb = L.Stats()
        self._chk(L.lib.lrnde_node_backward_recorded(self._ctx, B, _dev_ptr(du_end, "du_end", self.D), float(w_reg),
                                                     _dev_ptr(dx, "dx"), C.c_void_p(dp.data_ptr()), C.byref(sb)))
        return dict(dx=dx, dp=dp, stats_bwd=sb.asdict())

    def node_forward_record_ts(self, x, t0, t2, abstol, reltol, saveat, mode="unbiased", reg_type="error_estimate",
                               t1_or_rand=0.5, maxiters=1000, save_start=False, exact_pow=False):
        """the recorded layer forward with the layer's `saveat` kwarg: dict(u (nseries,B,D), t, u_end, reg_val, nfe, ...) —
        sol.u / sol.t as the caller of the layer sees them (after _CorrectedDESolution)"""
        B = x.numel() // self.D
        sv = np.ascontiguousarray(saveat, dtype=np.float32)
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        cap = int(sv.size) + 3 + (0 if sv.size or mode != "biased" else min(int(maxiters), 510))
        us = torch.empty((cap,) + tuple(x.shape), dtype=torch.float32, device=x.device)
        ts = np.empty(cap, dtype=np.float32)
        ns, reg, nfe, st, t1u = C.c_int32(), C.c_float(), C.c_int32(), L.Stats(), C.c_float()
        self._chk(L.lib.lrnde_node_forward_record_ts(
            self._ctx, _dev_ptr(x, "x", self.D), B, float(t0), float(t2), C.byref(o), L.MODE[mode], L.REG_TYPE[reg_type],
            float(t1_or_rand), sv.ctypes.data_as(C.POINTER(C.c_float)) if sv.size else None, int(sv.size), C.c_void_p(us.data_ptr()),
            ts.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(ns), C.byref(reg), C.byref(nfe), C.byref(st), C.byref(t1u)))
        n = int(ns.value)
        return dict(u=us[:n], t=ts[:n].copy(), u_end=us[n - 1], reg_val=np.float32(reg.value), nfe=int(nfe.value),
                    stats=st.asdict(), t1=np.float32(t1u.value))

    def node_backward_recorded_ts(self, du_series, w_reg=0.0):
        """pullback from the record of node_forward_record_ts for one cotangent per state of the series (nseries,B,D)"""
        du_series = du_series.contiguous()
        nser = int(du_series.shape[0])
        B = du_series[0].numel() // self.D
        dx = torch.empty_like(du_series[0])
        dp = torch.empty(int(L.lib.lrnde_param_count(C.byref(self.desc))), dtype=torch.float32, device=du_series.device)
        sb = L.Stats()
        self._chk(L.lib.lrnde_node_backward_recorded_ts(self._ctx, B, _dev_ptr(du_series, "du_series", self.D), nser, float(w_reg),
                                                        _dev_ptr(dx, "dx"), C.c_void_p(dp.data_ptr()), C.byref(sb)))
        return dict(dx=dx, dp=dp, stats_bwd=sb.asdict())

    def classifier_ce(self, u, pc, K, labels, want_grads=True):
        """Dense(D => K) + logitcrossentropy on device: dict(loss, logits, du, dpc)."""
        B = u.numel() // self.D
        if pc.numel() != K * (self.D + 1):
            raise ValueError(f"classifier parameters must have {K * (self.D + 1)} entries")
        if not (labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == B):
            raise ValueError("labels must be a CUDA int32 tensor of length B")
        logits = torch.empty((B, K), dtype=torch.float32, device=u.device)
        du = torch.empty_like(u) if want_grads else None
        dpc = torch.empty_like(pc) if want_grads else None
        loss = C.c_float()
        self._chk(L.lib.lrnde_classifier_ce(self._ctx, _dev_ptr(u, "u", self.D), B, _dev_ptr(pc, "pc"), int(K),
                                            C.c_void_p(labels.data_ptr()), C.byref(loss), C.c_void_p(logits.data_ptr()),
                                            C.c_void_p(du.data_ptr()) if want_grads else None,
                                            C.c_void_p(dpc.data_ptr()) if want_grads else None))
        return dict(loss=np.float32(loss.value), logits=logits, du=du, dpc=dpc)

    def node_forward_record_ce(self, x, t0, t2, abstol, reltol, pc, K, labels, mode="unbiased", reg_type="error_estimate",
                               t1_or_rand=0.5, maxiters=1000, save_start=False, exact_pow=False):
        """`node_forward_record` + `classifier_ce` on its sol.u[end] in one call (one synchronisation): (forward dict, head dict)"""
        B = x.numel() // self.D
        if pc.numel() != K * (self.D + 1):
            raise ValueError(f"classifier parameters must have {K * (self.D + 1)} entries")
        if not (labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == B):
            raise ValueError("labels must be a CUDA int32 tensor of length B")
        o = L.SolveOpts(float(abstol), float(reltol), int(maxiters), int(save_start), 0, int(exact_pow))
        u_end = torch.empty_like(x)
        logits = torch.empty((B, K), dtype=torch.float32, device=x.device)
        du = torch.empty_like(x)
        dpc = torch.empty_like(pc)
        reg, nfe, st, t1u, loss = C.c_float(), C.c_int32(), L.Stats(), C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_node_forward_record_ce(
            self._ctx, _dev_ptr(x, "x", self.D), B, float(t0), float(t2), C.byref(o), L.MODE[mode], L.REG_TYPE[reg_type],
            float(t1_or_rand), _dev_ptr(u_end, "u_end"), C.byref(reg), C.byref(nfe), C.byref(st), C.byref(t1u),
            _dev_ptr(pc, "pc"), int(K), C.c_void_p(labels.data_ptr()), C.byref(loss), C.c_void_p(logits.data_ptr()),
            C.c_void_p(du.data_ptr()), C.c_void_p(dpc.data_ptr())))
        fw = dict(u_end=u_end, reg_val=np.float32(reg.value), nfe=int(nfe.value), stats=st.asdict(), t1=np.float32(t1u.value))
        return fw, dict(loss=np.float32(loss.value), logits=logits, du=du, dpc=dpc)

    def bench_step(self, uprev, k1, t, dt, abstol, reltol, reps=50):
        """microseconds per launch of the full-step kernel (HIP events on the handle's stream)."""
        B = uprev.numel() // self.D
        us = C.c_float()
        self._chk(L.lib.lrnde_bench_step(self._ctx, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(k1, "k1", self.D), B,
                                         float(t), float(dt), float(abstol), float(reltol), int(reps), C.byref(us)))
        return float(us.value)

    def comm_count(self):
        """(nranks, kind) of the communicator the handle holds: kind 0 none, 1 RCCL, 2 in-process local (lrnde_comm_count)"""
        n, k = C.c_int32(), C.c_int32()
        self._chk(L.lib.lrnde_comm_count(self._ctx, C.byref(n), C.byref(k)))
        return int(n.value), int(k.value)

    def bench_exchange(self, B, reps=100):
        """microseconds per per-step exchange (collective: every rank calls it); 0.0 on an unsharded handle"""
        us = C.c_float()
        self._chk(L.lib.lrnde_bench_exchange(self._ctx, int(B), int(reps), C.byref(us)))
        return float(us.value)

    def set_reports(self, on):
        """diagnostic (lrnde_hooks.h): False makes the solve loop poll by copies instead of reading the per-launch reports"""
        self._chk(L.lib.lrnde_set_reports(self._ctx, 1 if on else 0))

    def set_overlap(self, on):
        """diagnostic (lrnde_hooks.h): False keeps the layer forward's local step / regulariser sweep on the handle's own stream"""
        self._chk(L.lib.lrnde_set_overlap(self._ctx, 1 if on else 0))

    def last_solve_kernel_ms(self):
        ms, n = C.c_float(), C.c_int32()
        L.lib.lrnde_last_solve_kernel_ms(self._ctx, C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)


def biased_index(r01, m):
    """index into sol.t[1:end-1] (m entries) from one uniform draw: (int)(r01 * (float)m) in float32, clamped — the
    arithmetic of lrnde_node_forward (csrc/lrnde_kernels.hip)"""
    return max(0, min(int(np.float32(r01) * np.float32(m)), m - 1))


# ----- solution object: the fields the reference reads (src/utils.jl:7-9,25-46) -----
class ODESolution:
    def __init__(self, u, t, nf, naccept=0, nreject=0, retcode="Success"):
        self.u, self.t = list(u), [np.float32(v) for v in t]
        self.destats = SimpleNamespace(nf=int(nf), naccept=int(naccept), nreject=int(nreject))
        self.retcode = retcode

    def __call__(self, t):  # saveat-only solutions answer at their knots (neural_ode.jl:34)
        for ti, ui in zip(self.t, self.u):
            if ti == np.float32(t):
                return ui
        raise ValueError("this solution only stores its saveat knots")


def diffeqsol_to_array(sol):
    """src/utils.jl:37-40."""
    return sol.u[-1] if isinstance(sol, ODESolution) else sol


def diffeqsol_to_timeseries(sol):
    """src/utils.jl:42-46: states stacked along a new second-to-last (Julia) dimension."""
    return torch.stack(list(sol.u), dim=0)


class NeuralODE:
    """src/layers/neural_ode.jl:1-116.  `(sol, st) = node(x, ps, st)`."""

    def __init__(self, model, *, solver="Tsit5", sensealg=None, tspan=(0.0, 1.0), regularize=True,
                 maxiters=1000, regularize_type="error_estimate", **kwargs):
        if isinstance(regularize, bool):  # :14-16
            regularize = "unbiased" if regularize else "none"
        regularize, regularize_type = _sym(regularize), _sym(regularize_type)
        _check_valid_regularize(regularize)
        _check_valid_regularize(regularize_type, _VALID_REG_TYPES)
        self.model, self.solver, self.sensealg = model, _solver_name(solver), sensealg
        self.tspan = (np.float32(tspan[0]), np.float32(tspan[1]))
        self.maxiters, self.kwargs = int(maxiters), dict(kwargs)
        self.regularize, self.regularize_type = regularize, regularize_type
        from .conv import conv_topology
        self._conv = conv_topology(model)  # (C, Hc, act, eps) for the CIFAR node_core, else None
        self.desc = None if self._conv else _mlp_desc(model)
        if self._conv and self.solver != "tsit5":
            raise NotImplementedError("VCAB3 / VCABM3 are built for the MLP field's handle (csrc/lrnde_adams.hpp)")
        self._handle = None
        self._bound = False

    def initialstates(self, rng):
        """:27-31 — burns one normal draw, then replicates the rng."""
        rng.standard_normal()
        return dict(model={}, nfe=-1, reg_val=np.float32(0.0), rng=copy.deepcopy(rng), training=True)

    def handle(self, x=None):
        if self._conv:  # the conv field's handle is per image size: x is (B, C, H, W)
            from .conv import ConvHandle
            if x is None and self._handle is None:
                raise ValueError("the conv field needs the input to size its handle")
            if x is not None:
                if x.dim() != 4 or x.shape[1] != self._conv[0]:
                    raise ValueError(f"conv field input must be (B, {self._conv[0]}, H, W), got {tuple(x.shape)}")
                hw = (int(x.shape[3]), int(x.shape[2]))
                if self._handle is None or self._hw != hw:
                    Cst, Hc, act, eps = self._conv
                    self._handle = ConvHandle(hw[0], hw[1], Cst, Hc, act=act, bn_train=True, bn_eps=eps,
                                              compute_dtype=self.kwargs.get("compute_dtype", "f32"))
                    self._hw, self._bound = hw, False
            return self._handle
        if self._handle is None:
            self._handle = Handle(self.desc)
            self._handle.set_solver(self.solver)
        return self._handle

    def _bind(self, ps, x=None, changed=True):
        """Hands `ps` to the device handle.  The parameters are repacked on EVERY call unless the caller states
        `changed=False` (same values as the last bind): equality is never inferred from a pointer or a torch version
        counter — an in-place numpy / ctypes / foreign-kernel update of the same storage bumps neither, and a freed
        buffer can come back at the same address.  The repack is four small kernels (a few microseconds)."""
        h = self.handle(x)
        if changed or not self._bound:
            h.set_params(ps)
            self._bound = True
        return h

    def _model_state_in(self, h, st):
        """conv field: the BatchNorm running statistics are the layer's model state (src/layers/neural_ode.jl:44-48)"""
        if self._conv:
            if isinstance(st.get("model"), dict) and st["model"].get("bn_state") is not None:
                h.set_bn_state(st["model"]["bn_state"])
            h.set_bn_mode(bool(st["training"]))  # Lux.testmode flips every nested layer: running statistics, no update

    def _model_state_out(self, h, st):
        if self._conv:
            return dict(st["model"] if isinstance(st.get("model"), dict) else {}, bn_state=h.get_bn_state())
        return st["model"]

    def __call__(self, x, ps, st):
        h = self._bind(ps, x)
        self._model_state_in(h, st)
        t0, t2 = self.tspan
        kw = self.kwargs
        abstol, reltol = kw.get("abstol", 1e-6), kw.get("reltol", 1e-3)  # OrdinaryDiffEq defaults
        save_start = kw.get("save_start", True)
        saveat = kw.get("saveat", None)
        mode = self.regularize if st["training"] else "none"  # :62-66,86
        common = dict(maxiters=self.maxiters, save_start=save_start)
        if mode == "none":  # _vanilla_node_fallback :56-60
            sv = [t2] if saveat is None else list(saveat)
            r = h.solve(x, t0, t2, abstol, reltol, saveat=sv, save_everystep=False, **common)
            sol = ODESolution(r["u"], r["t"], r["stats"]["nf"], r["stats"]["naccept"], r["stats"]["nreject"])
            return sol, dict(model=self._model_state_out(h, st), nfe=r["stats"]["nf"], reg_val=np.float32(0.0), rng=st["rng"],
                             training=st["training"])
        rng = copy.deepcopy(st["rng"])  # Lux.replicate(st.rng)
        if mode == "unbiased":  # :68-84
            t1 = np.float32(rng.random(dtype=np.float32) * (t2 - t0) + t0)
            needs_correction = saveat is not None
            sv = sorted(list(saveat) + [t1]) if needs_correction else [t1, t2]
            r = h.solve(x, t0, t2, abstol, reltol, saveat=sv, save_everystep=False, **common)
            ts = list(r["t"])
            i1 = max(i for i, tv in enumerate(ts) if tv == t1)
            u1 = r["u"][i1]
        else:  # :88-100
            r = h.solve(x, t0, t2, abstol, reltol, saveat=() if saveat is None else saveat,
                        save_everystep=saveat is None, cap=min(self.maxiters, 510) + 2, **common)
            ts = list(r["t"])
            needs_correction = False
            # rand(rng, sol.t[1:end-1]) with the ONE draw convention of this package (pullback, run_training_step and the
            # C side's lrnde_node_forward use the same): one uniform float32 r in [0,1), index floor(r * m)
            r01 = np.float32(rng.random(dtype=np.float32))
            i1 = biased_index(r01, len(ts) - 1)
            t1, u1 = ts[i1], r["u"][i1]
        model_state = self._model_state_out(h, st)  # as it is when the solve returns (:52); the local step leaves no trace
        # _get_ode_integrator :33-38 + _perform_step :77
        dt, k1 = h.init_dt(u1.contiguous(), t1, t2, abstol, reltol)
        ps_out = h.perform_step(u1.contiguous(), k1, t1, dt, abstol, reltol)
        reg_val = ps_out["reg_stiff"] if self.regularize_type == "stiffness_estimate" else ps_out["reg_error"]
        nfe = r["stats"]["nf"] + 6 + 3  # :79 with src/perform_step.jl:31
        us, tt = list(r["u"]), ts
        if needs_correction:  # _CorrectedDESolution, src/utils.jl:31-33
            keep = [i for i, tv in enumerate(tt) if tv != t1]
            us, tt = [us[i] for i in keep], [tt[i] for i in keep]
        sol = ODESolution(us, tt, r["stats"]["nf"], r["stats"]["naccept"], r["stats"]["nreject"])
        if self._conv:
            h.set_bn_state(model_state["bn_state"])
        return sol, dict(model=model_state, nfe=nfe, reg_val=reg_val, rng=rng, training=st["training"])

    def pullback(self, x, ps, st, du, w_reg=0.0):
        """What `Zygote.pullback` returns for this layer in the reference's training steps
        (experiments/src/utils.jl:104-115) for  loss = <du, sol> + w_reg * reg_val: (dx, dps, info).
        du: the cotangent of sol.u[end] (what `diffeqsol_to_array` consumers send back), or — with a user `saveat` — one
        cotangent per state of the returned solution, stacked (nseries, B, D) (`diffeqsol_to_timeseries` consumers,
        src/utils.jl:42-46, experiments/src/construct.jl:244-249).  One forward with the dense record, then the backward
        from it; t1 comes from the same single draw of st['rng'] as in `__call__` (biased_index), so info['reg_val'] /
        info['t1'] are `__call__`'s."""
        h = self._bind(ps, x)
        t0, t2 = self.tspan
        kw = self.kwargs
        abstol, reltol = kw.get("abstol", 1e-6), kw.get("reltol", 1e-3)
        mode = self.regularize if st["training"] else "none"
        rng = copy.deepcopy(st["rng"])
        r01 = np.float32(rng.random(dtype=np.float32))
        t1_or_rand = np.float32(r01 * (t2 - t0) + t0) if mode == "unbiased" else r01
        common = dict(mode=mode, reg_type=self.regularize_type, t1_or_rand=t1_or_rand, maxiters=self.maxiters,
                      save_start=kw.get("save_start", True))
        saveat = kw.get("saveat", None)
        if saveat is not None:
            if self._conv:
                raise NotImplementedError("time-series cotangents are built for the MLP field")
            fw = h.node_forward_record_ts(x, t0, t2, abstol, reltol, saveat, **common)
            nser = int(fw["u"].shape[0])
            if du.dim() == x.dim():  # a cotangent for sol.u[end] only
                full = torch.zeros_like(fw["u"])
                full[nser - 1] = du
                du = full
            if int(du.shape[0]) != nser:
                raise ValueError(f"{int(du.shape[0])} cotangents for a solution of {nser} saved states")
            bw = h.node_backward_recorded_ts(du, w_reg=w_reg)
        else:
            fw = h.node_forward_record(x, t0, t2, abstol, reltol, **common)
            bw = h.node_backward_recorded(du, w_reg=w_reg) if not self._conv else h.node_backward_recorded(x.shape[0], du, w_reg=w_reg)
        info = dict(bw, reg_val=fw["reg_val"], t1=fw["t1"], nfe=fw["nfe"], u_end=fw["u_end"], stats_fwd=fw["stats"],
                    sol_t=fw.get("t"), sol_u=fw.get("u"))
        return bw["dx"], bw["dp"], info
