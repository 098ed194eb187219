"""NeuralDSDE mirror (src/layers/neural_sde.jl) on the liblrnde SDE entry points.

The layer runs what the reference's runs (src/layers/neural_sde.jl:50-123): an ADAPTIVE solve — the Lamba Euler-Heun step
(`_perform_step(::LambaEulerHeunConstantCache)`, src/perform_step.jl:172-206) under a PI controller on its error estimate,
on a Brownian path drawn up front on a uniform grid (`lrnde_sde_node_forward_record`) — the local step at (sol(t1), t1) for
reg_val, user `saveat` with the `_CorrectedDESolution` filter, and the pullback as the reverse sweep over the recorded
accepted steps (`lrnde_sde_node_backward_recorded`).  `adaptive=False` keeps the fixed-grid integrator of rounds 1-2 (also the
only mode of the Milstein and four-stage SRI steps, :108-170 and :49-106).  SOSRI's tableau and its RSWM noise process live in
un-vendored StochasticDiffEq and are not restated: `solver="SOSRI"` raises, `solver="SRI"` takes the tableau from the caller.
"""
import copy
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .layers import (Chain, Dense, ODESolution, _check_valid_regularize, _dev_ptr, _mlp_desc, _sym)


class SdeHandle:
    def __init__(self, drift_desc, diffusion_bias=True, device=None, stream=None):
        if not torch.cuda.is_available():
            raise RuntimeError("liblrnde needs a GPU (gfx950); there is no CPU fallback")
        self.D = drift_desc.state_dim
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._stream = torch.cuda.current_stream(self.device) if stream is None else stream
        self._h = C.c_void_p()
        rc = L.lib.lrnde_sde_create(C.byref(self._h), C.byref(drift_desc), int(bool(diffusion_bias)), self.device,
                                    C.c_void_p(self._stream.cuda_stream))
        if rc != 0:
            raise L.LrndeError(rc, "lrnde_sde_create failed")
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            L.lib.lrnde_sde_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = L.lib.lrnde_sde_last_error(self._h)
            raise L.LrndeError(rc, msg.decode() if msg else "")

    def set_params(self, p_drift, p_diffusion):
        dev = f"cuda:{self.device}"
        pd = torch.as_tensor(p_drift, dtype=torch.float32).to(dev).contiguous().reshape(-1)
        pg = torch.as_tensor(p_diffusion, dtype=torch.float32).to(dev).contiguous().reshape(-1)
        self._keep = (pd, pg)
        self._chk(L.lib.lrnde_sde_set_params(self._h, C.c_void_p(pd.data_ptr()), pd.numel(),
                                             C.c_void_p(pg.data_ptr()), pg.numel()))

    def euler_heun_step(self, uprev, dW, t, dt, abstol, reltol, delta):
        B = uprev.numel() // self.D
        u = torch.empty_like(uprev)
        ee, rv = C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_sde_euler_heun_step(self._h, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D),
                                                  B, float(t), float(dt), float(abstol), float(reltol), float(delta),
                                                  _dev_ptr(u, "u"), C.byref(ee), C.byref(rv)))
        return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))

    def rkmil_step(self, uprev, dW, t, dt, abstol, reltol):
        """`_perform_step(integrator, ::RKMilCommuteConstantCache, p)` (src/perform_step.jl:108-170), diagonal noise, Ito."""
        B = uprev.numel() // self.D
        u = torch.empty_like(uprev)
        ee, rv = C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_sde_rkmil_step(self._h, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D), B,
                                             float(t), float(dt), float(abstol), float(reltol), _dev_ptr(u, "u"),
                                             C.byref(ee), C.byref(rv)))
        return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))


    def sri_step(self, tableau, uprev, dW, dZ, t, dt, abstol, reltol, delta):
        """`_perform_step(integrator, ::FourStageSRIConstantCache, p)` (src/perform_step.jl:49-106), diagonal noise;
        tableau: dict / sequence with the 51 coefficients named as the reference unpacks them (L.SRI_FIELDS) — SOSRI's
        values live in StochasticDiffEq and are the caller's to supply."""
        tab = L.SriTableau(*[float(tableau[k]) for k in L.SRI_FIELDS]) if isinstance(tableau, dict) else L.SriTableau(*[float(v) for v in tableau])
        B = uprev.numel() // self.D
        u = torch.empty_like(uprev)
        ee, rv = C.c_float(), C.c_float()
        self._chk(L.lib.lrnde_sde_sri_step(self._h, C.byref(tab), _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D),
                                           _dev_ptr(dZ, "dZ", self.D), B, float(t), float(dt), float(abstol), float(reltol),
                                           float(delta), _dev_ptr(u, "u"), C.byref(ee), C.byref(rv)))
        return dict(u=u, eest=np.float32(ee.value), reg_val=np.float32(rv.value))

    def solve_fixed(self, u0, dW, t0, dt, abstol, reltol, delta=1.0 / 6.0, solver="EulerHeun"):
        """nsteps = dW.shape[0] steps on a fixed grid in one call (no host round trip per step): dict(u (nsteps,B,D),
        eest (nsteps,), reg_val (nsteps,)); step i equals the single-step call from t0 + i*dt"""
        nsteps = int(dW.shape[0])
        B = u0.numel() // self.D
        dW = dW.contiguous()
        u = torch.empty((nsteps,) + tuple(u0.shape), dtype=torch.float32, device=u0.device)
        ee = (C.c_float * nsteps)(); rv = (C.c_float * nsteps)()
        self._chk(L.lib.lrnde_sde_solve_fixed(self._h, 1 if solver.startswith("RKMil") else 0, _dev_ptr(u0, "u0", self.D),
                                              _dev_ptr(dW, "dW"), B, float(t0), float(dt), nsteps, float(abstol), float(reltol),
                                              float(delta), _dev_ptr(u, "u"), ee, rv))
        return dict(u=u, eest=np.frombuffer(ee, dtype=np.float32).copy(), reg_val=np.frombuffer(rv, dtype=np.float32).copy())


    def solve_adaptive(self, u0, W, t0, t1, abstol, reltol, delta=1.0 / 6.0, dt0=None, gamma=0.9, qmin=0.2, qmax=1.125,
                       beta1=7.0 / 50.0, beta2=2.0 / 25.0, maxiters=10000):
        """adaptive Euler-Heun on the caller's Brownian path W ((nfine+1, B, D), W[0] = 0, uniform grid over (t0, t1)):
        dict(u_end, stats, trace) — steps are whole grid intervals, EEst drives a PI controller (lrnde.h)"""
        nfine = int(W.shape[0]) - 1
        B = u0.numel() // self.D
        o = L.SdeAdaptOpts(float(abstol), float(reltol), float(delta), float((t1 - t0) / nfine if dt0 is None else dt0),
                           float(gamma), float(qmin), float(qmax), float(beta1), float(beta2), int(maxiters))
        u_end = torch.empty_like(u0)
        st = L.Stats()
        ntr = int(maxiters) + 4
        tr = (L.TraceRow * ntr)()
        self._chk(L.lib.lrnde_sde_solve_adaptive(self._h, _dev_ptr(u0, "u0", self.D), _dev_ptr(W.contiguous(), "W"), nfine, B,
                                                 float(t0), float(t1), C.byref(o), _dev_ptr(u_end, "u_end"), C.byref(st), tr, ntr))
        nt = st.naccept + st.nreject
        trace = np.array([(tr[i].t, tr[i].dt, tr[i].eest, tr[i].accepted) for i in range(nt)],
                         dtype=[("t", "f4"), ("dt", "f4"), ("eest", "f4"), ("accepted", "i4")])
        return dict(u_end=u_end, stats=st.asdict(), trace=trace)

    def _pcounts(self):
        return self._keep[0].numel(), self._keep[1].numel()

    def node_forward_record(self, x, W, t0, t2, abstol, reltol, mode="unbiased", t1_or_rand=0.5, z_local=None, saveat=(),
                            save_start=-1, delta=1.0 / 6.0, dt0=0.0, gamma=0.9, qmin=0.2, qmax=1.125, beta1=7.0 / 50.0,
                            beta2=2.0 / 25.0, maxiters=10000):
        """the NeuralDSDE layer forward (lrnde_sde_node_forward_record): adaptive solve on the path W ((nfine+1, B, D), W[0] = 0),
        sol.u / sol.t as the layer's caller sees them, reg_val of the local step; keeps the record for one
        `node_backward_recorded`.  dt0 = 0: automatic initial dt."""
        nfine = int(W.shape[0]) - 1
        B = x.numel() // self.D
        W = W.contiguous()
        sv = np.ascontiguousarray(saveat, dtype=np.float32)
        o = L.SdeAdaptOpts(float(abstol), float(reltol), float(delta), float(dt0), float(gamma), float(qmin), float(qmax),
                           float(beta1), float(beta2), int(maxiters))
        cap = int(sv.size) + 3 + (nfine + 1 if (mode == "biased" and not sv.size) else 0)
        us = torch.empty((cap,) + tuple(x.shape), dtype=torch.float32, device=x.device)
        ts = np.empty(cap, dtype=np.float32)
        ns, reg, nf, ng, st, t1u = C.c_int32(), C.c_float(), C.c_int32(), C.c_int32(), L.Stats(), C.c_float()
        if z_local is not None:
            z_local = z_local.contiguous()
        self._chk(L.lib.lrnde_sde_node_forward_record(
            self._h, _dev_ptr(x, "x", self.D), _dev_ptr(W, "W"), nfine, B, float(t0), float(t2), C.byref(o), L.MODE[mode],
            float(t1_or_rand), None if z_local is None else _dev_ptr(z_local, "z_local", self.D), int(save_start),
            sv.ctypes.data_as(C.POINTER(C.c_float)) if sv.size else None, int(sv.size), C.c_void_p(us.data_ptr()),
            ts.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(ns), C.byref(reg), C.byref(nf), C.byref(ng), C.byref(st), C.byref(t1u)))
        self._node_keep = (W, x)   # the record refers to the caller's path: keep it alive until the backward
        n = int(ns.value)
        return dict(u=us[:n], t=ts[:n].copy(), u_end=us[n - 1], reg_val=np.float32(reg.value), nfe_drift=int(nf.value),
                    nfe_diffusion=int(ng.value), stats=st.asdict(), t1=np.float32(t1u.value))

    def node_backward_recorded(self, du_series, w_reg=0.0):
        """pullback of sum_j <du_j, sol.u[j]> + w_reg * reg_val from the record of `node_forward_record`"""
        du_series = du_series.contiguous()
        nser = int(du_series.shape[0])
        B = du_series[0].numel() // self.D
        nf, ng = self._pcounts()
        dx = torch.empty_like(du_series[0])
        dpf = torch.empty(nf, dtype=torch.float32, device=dx.device)
        dpg = torch.empty(ng, dtype=torch.float32, device=dx.device)
        self._chk(L.lib.lrnde_sde_node_backward_recorded(self._h, B, _dev_ptr(du_series, "du_series", self.D), nser, float(w_reg),
                                                         _dev_ptr(dx, "dx"), C.c_void_p(dpf.data_ptr()), C.c_void_p(dpg.data_ptr())))
        return dict(dx=dx, dp_drift=dpf, dp_diff=dpg)

    def solve_fixed_backward(self, u0, u_traj, dW, t0, dt, du_end, solver="EulerHeun"):
        """pullback of solve_fixed (Euler-Heun, or solver="RKMil": the Milstein step) for <du_end, u_traj[-1]>:
        dict(dx, dp_drift, dp_diff)"""
        nsteps = int(dW.shape[0])
        B = u0.numel() // self.D
        nf, ng = self._pcounts()
        dx = torch.empty_like(u0)
        dpf = torch.empty(nf, dtype=torch.float32, device=u0.device)
        dpg = torch.empty(ng, dtype=torch.float32, device=u0.device)
        fn = L.lib.lrnde_sde_solve_fixed_backward_rkmil if solver.startswith("RKMil") else L.lib.lrnde_sde_solve_fixed_backward
        self._chk(fn(self._h, _dev_ptr(u0, "u0", self.D), _dev_ptr(u_traj.contiguous(), "u_traj"),
                                                       _dev_ptr(dW.contiguous(), "dW"), B, float(t0), float(dt), nsteps,
                                                       _dev_ptr(du_end, "du_end", self.D), _dev_ptr(dx, "dx"),
                                                       C.c_void_p(dpf.data_ptr()), C.c_void_p(dpg.data_ptr())))
        return dict(dx=dx, dp_drift=dpf, dp_diff=dpg)

    def sri_step_backward(self, tableau, uprev, dW, dZ, t, dt, abstol, reltol, delta, du_new=None, w_reg=0.0, want_dx=True,
                          dp_drift=None, dp_diff=None):
        """reverse sweep of one four-stage SRI step (lrnde_sde_sri_step_backward) for <du_new, u'> + w_reg * EEst*dt:
        dict(dx, dp_drift, dp_diff, reg_val); dp_* are accumulated into the given tensors (fresh zeros by default)"""
        B = uprev.numel() // self.D
        tab = tableau if isinstance(tableau, L.SriTableau) else L.SriTableau(*[float(v) for v in tableau])
        nf, ng = self._pcounts()
        dpf = torch.zeros(nf, dtype=torch.float32, device=uprev.device) if dp_drift is None else dp_drift
        dpg = torch.zeros(ng, dtype=torch.float32, device=uprev.device) if dp_diff is None else dp_diff
        dx = torch.empty_like(uprev) if want_dx else None
        rv = C.c_float()
        self._chk(L.lib.lrnde_sde_sri_step_backward(
            self._h, C.byref(tab), _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D), _dev_ptr(dZ, "dZ", self.D), B, float(t),
            float(dt), float(abstol), float(reltol), float(delta), None if du_new is None else _dev_ptr(du_new.contiguous(), "du_new", self.D),
            float(w_reg), None if dx is None else _dev_ptr(dx, "dx"), C.c_void_p(dpf.data_ptr()), C.c_void_p(dpg.data_ptr()), C.byref(rv)))
        return dict(dx=dx, dp_drift=dpf, dp_diff=dpg, reg_val=np.float32(rv.value))

    def rkmil_reg_grad(self, uprev, dW, t, dt, abstol, reltol):
        """d (EEst*dt) / d (p_drift, p_diffusion) of one local Milstein step (src/perform_step.jl:108-170), uprev / dW / dt constant"""
        B = uprev.numel() // self.D
        nf, ng = self._pcounts()
        dpf = torch.empty(nf, dtype=torch.float32, device=uprev.device)
        dpg = torch.empty(ng, dtype=torch.float32, device=uprev.device)
        rv = C.c_float()
        self._chk(L.lib.lrnde_sde_rkmil_reg_grad(self._h, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D), B, float(t), float(dt),
                                                 float(abstol), float(reltol), C.c_void_p(dpf.data_ptr()), C.c_void_p(dpg.data_ptr()), C.byref(rv)))
        return dict(dp_drift=dpf, dp_diff=dpg, reg_val=np.float32(rv.value))

    def euler_heun_reg_grad(self, uprev, dW, t, dt, abstol, reltol, delta):
        """d (EEst*dt) / d (p_drift, p_diffusion) of one local Euler-Heun step, uprev / dW / dt constant"""
        B = uprev.numel() // self.D
        nf, ng = self._pcounts()
        dpf = torch.empty(nf, dtype=torch.float32, device=uprev.device)
        dpg = torch.empty(ng, dtype=torch.float32, device=uprev.device)
        rv = C.c_float()
        self._chk(L.lib.lrnde_sde_euler_heun_reg_grad(self._h, _dev_ptr(uprev, "uprev", self.D), _dev_ptr(dW, "dW", self.D), B,
                                                      float(t), float(dt), float(abstol), float(reltol), float(delta),
                                                      C.c_void_p(dpf.data_ptr()), C.c_void_p(dpg.data_ptr()), C.byref(rv)))
        return dict(dp_drift=dpf, dp_diff=dpg, reg_val=np.float32(rv.value))


class NeuralDSDE:
    """`(sol, st) = nsde(x, ps, st)`; ps = dict(drift=flat, diffusion=[vec(Wg); bg]).
    src/layers/neural_sde.jl:1-123.  Default (solver="EulerHeun", adaptive=True): the ADAPTIVE solve on a Brownian path of
    `nfine` grid intervals drawn from st["rng"] (or given as `noise=` / `path=`), kwargs `abstol`, `reltol`, `saveat`, `save_start`
    as the reference's; `adaptive=False` (and the Milstein / SRI steps): the fixed grid of `nsteps` steps."""

    def __init__(self, drift, diffusion, *, solver="EulerHeun", sensealg=None, tspan=(0.0, 1.0),
                 regularize="unbiased", maxiters=1000, nsteps=20, delta=1.0 / 6.0, tableau=None, adaptive=None, nfine=256,
                 dt0=0.0, **kwargs):
        regularize = _sym(regularize)
        _check_valid_regularize(regularize)
        if solver in ("SRI", "FourStageSRI"):
            if tableau is None:
                raise ValueError("solver='SRI' (src/perform_step.jl:49-106) needs tableau=: the 51 coefficients of the "
                                 "FourStageSRIConstantCache (SOSRI's live in un-vendored StochasticDiffEq)")
        elif solver not in ("EulerHeun", "LambaEulerHeun", "RKMil", "RKMilCommute"):
            raise NotImplementedError("on the device: the Euler-Heun step (src/perform_step.jl:172-206), the Milstein step "
                                      "(:108-170) and the four-stage SRI step (:49-106, solver='SRI' with tableau=); SOSRI's own "
                                      "tableau and adaptive noise process live in un-vendored StochasticDiffEq")
        self.solver = "SRI" if solver in ("SRI", "FourStageSRI") else ("RKMil" if solver.startswith("RKMil") else "EulerHeun")
        self.tableau = tableau
        if not isinstance(diffusion, Dense) or diffusion.in_dims != diffusion.out_dims:
            raise NotImplementedError("diffusion must be Dense(D => D) (experiments/src/construct.jl:205)")
        self.drift, self.diffusion = drift, diffusion
        self.desc = _mlp_desc(drift if isinstance(drift, Chain) else drift)
        if self.desc.state_dim != diffusion.in_dims:
            raise ValueError("drift and diffusion state sizes differ")
        self.tspan = (np.float32(tspan[0]), np.float32(tspan[1]))
        self.regularize, self.maxiters, self.nsteps, self.delta = regularize, int(maxiters), int(nsteps), float(delta)
        self.adaptive = (self.solver == "EulerHeun") if adaptive is None else bool(adaptive)
        if self.adaptive and self.solver != "EulerHeun":
            raise NotImplementedError("the adaptive solve is built on the Euler-Heun step (src/perform_step.jl:172-206)")
        self.nfine, self.dt0 = int(nfine), float(dt0)
        self.kwargs = dict(kwargs)
        self._handle = None
        self._last_adaptive = None

    def initialstates(self, rng):
        rng.standard_normal()  # :23
        return dict(drift={}, diffusion={}, nfe_drift=-1, nfe_diffusion=-1, reg_val=np.float32(0.0),
                    rng=copy.deepcopy(rng), training=True)

    def handle(self):
        if self._handle is None:
            self._handle = SdeHandle(self.desc)
        return self._handle

    def _draw_path(self, rng, shape, device):
        """W on the uniform grid (nfine + 1, B, D), W[0] = 0, and the local step's standard-normal draw, from the host stream"""
        t0, t2 = self.tspan
        h = np.float32((t2 - t0) / np.float32(self.nfine))
        inc = rng.standard_normal((self.nfine,) + tuple(shape)).astype(np.float32) * np.float32(np.sqrt(h))
        W = np.concatenate([np.zeros((1,) + tuple(shape), np.float32), np.cumsum(inc, axis=0, dtype=np.float32)], axis=0)
        z = rng.standard_normal(tuple(shape)).astype(np.float32)
        return torch.from_numpy(W).to(device), torch.from_numpy(z).to(device)

    def _call_adaptive(self, x, ps, st, path=None, z_local=None):
        h = self.handle()
        h.set_params(ps["drift"], ps["diffusion"])
        t0, t2 = self.tspan
        abstol, reltol = self.kwargs.get("abstol", 1e-2), self.kwargs.get("reltol", 1e-2)
        rng = copy.deepcopy(st["rng"])
        if path is None:
            path, z_draw = self._draw_path(rng, x.shape, x.device)
            z_local = z_draw if z_local is None else z_local
        else:
            path = torch.as_tensor(path, dtype=torch.float32).to(x.device)
            if z_local is None:
                z_local = torch.from_numpy(rng.standard_normal(tuple(x.shape)).astype(np.float32)).to(x.device)
        mode = self.regularize if st["training"] else "none"
        r01 = np.float32(rng.random(dtype=np.float32)) if mode != "none" else np.float32(0)
        t1_or_rand = np.float32(r01 * (t2 - t0) + t0) if mode == "unbiased" else r01     # :92 / :114
        saveat = self.kwargs.get("saveat", ())
        r = h.node_forward_record(x, path, t0, t2, abstol, reltol, mode=mode, t1_or_rand=float(t1_or_rand), z_local=z_local,
                                  saveat=() if saveat is None else saveat, save_start=int(self.kwargs.get("save_start", -1)),
                                  delta=self.delta, dt0=self.dt0, maxiters=self.maxiters)
        self._last_adaptive = dict(nseries=int(r["u"].shape[0]), mode=mode)
        sol = ODESolution([r["u"][i] for i in range(r["u"].shape[0])], [np.float32(t) for t in r["t"]], r["nfe_drift"])
        sol.stats = r["stats"]
        return sol, dict(drift=st["drift"], diffusion=st["diffusion"], nfe_drift=r["nfe_drift"], nfe_diffusion=r["nfe_diffusion"],
                         reg_val=r["reg_val"], rng=rng, training=st["training"])

    def __call__(self, x, ps, st, noise=None, path=None, z_local=None):
        if self.adaptive:
            return self._call_adaptive(x, ps, st, path=path if path is not None else noise, z_local=z_local)
        h = self.handle()
        h.set_params(ps["drift"], ps["diffusion"])
        t0, t2 = self.tspan
        abstol, reltol = self.kwargs.get("abstol", 1e-2), self.kwargs.get("reltol", 1e-2)
        n = self.nsteps
        dt = np.float32((t2 - t0) / np.float32(n))
        rng = copy.deepcopy(st["rng"])
        if noise is None:  # W.dW ~ sqrt(dt) N(0,1), drawn on the host stream
            noise = (rng.standard_normal((n + 1,) + tuple(x.shape)).astype(np.float32) * np.float32(np.sqrt(dt)))
        noise = torch.as_tensor(noise, dtype=torch.float32).to(x.device)
        # step(u, i, t): one step with the i-th increments of the noise process
        if self.solver == "SRI":  # second increment dZ (W.dZ), drawn after dW from the same host stream
            dz = torch.as_tensor(rng.standard_normal((n + 1,) + tuple(x.shape)).astype(np.float32) * np.float32(np.sqrt(dt))).to(x.device)
            step = lambda uu, i, tt: h.sri_step(self.tableau, uu, noise[i].contiguous(), dz[i].contiguous(), tt, dt, abstol, reltol, self.delta)
            us, u = [], x
            for i in range(n):
                u = step(u, i, np.float32(t0 + np.float32(i) * dt))["u"]
                us.append(u)
            self._last_solve = dict(us=us, dW=noise, dZ=dz, t0=t0, dt=dt, abstol=abstol, reltol=reltol)
        else:
            if self.solver == "RKMil":
                step = lambda uu, i, tt: h.rkmil_step(uu, noise[i].contiguous(), tt, dt, abstol, reltol)
            else:
                step = lambda uu, i, tt: h.euler_heun_step(uu, noise[i].contiguous(), tt, dt, abstol, reltol, self.delta)
            traj = h.solve_fixed(x, noise[:n], t0, dt, abstol, reltol, self.delta, self.solver)  # the n steps, one host sync
            us = [traj["u"][i] for i in range(n)]
            self._last_solve = dict(u_traj=traj["u"], dW=noise[:n], t0=t0, dt=dt)
        ts = [np.float32(t0 + np.float32(i + 1) * dt) if i + 1 < n else t2 for i in range(n)]
        per_step = {"RKMil": (1, 2), "SRI": (4, 4)}.get(self.solver, (3, 3))  # (drift, diffusion) evaluations of one step
        nfe, nfe_g = per_step[0] * n, per_step[1] * n
        mode = self.regularize if st["training"] else "none"
        reg_val = np.float32(0.0)
        if mode != "none":
            if mode == "unbiased":  # :88-105: t1 uniform in (t0,t2); sol(t1) by linear interpolation
                t1 = np.float32(rng.random(dtype=np.float32) * (t2 - t0) + t0)
                j = min(int((t1 - t0) / dt), n - 1)
                ta = t0 if j == 0 else ts[j - 1]
                ua = x if j == 0 else us[j - 1]
                th = np.float32((t1 - ta) / (ts[j] - ta))
                u1 = (ua + th * (us[j] - ua)).contiguous()
            else:  # :109-123: rand(rng, sol.t[1:end-1]) — every saved time but the last, t0 included (the SDE solve saves
                # its start); one uniform draw, index floor(r * m) (layers.biased_index: the package's one convention)
                from .layers import biased_index
                j = biased_index(np.float32(rng.random(dtype=np.float32)), n)
                t1, u1 = (t0, x) if j == 0 else (ts[j - 1], us[j - 1])
            # the local step's integrator is a fresh `init` on (t1, t2) (:96,116): its first dt is clipped to the span
            dt_loc = np.float32(min(dt, np.float32(t2 - t1)))
            if self.solver == "SRI":
                r = h.sri_step(self.tableau, u1, noise[n].contiguous(), dz[n].contiguous(), t1, dt_loc, abstol, reltol, self.delta)
            elif self.solver == "RKMil":
                r = h.rkmil_step(u1, noise[n].contiguous(), t1, dt_loc, abstol, reltol)
            else:
                r = h.euler_heun_step(u1, noise[n].contiguous(), t1, dt_loc, abstol, reltol, self.delta)
            reg_val = r["reg_val"]
            nfe += per_step[0]; nfe_g += per_step[1]
            self._last_local = dict(t1=t1, dt=dt_loc, u1=u1, dW=noise[n].contiguous(), abstol=abstol, reltol=reltol,
                                    dZ=dz[n].contiguous() if self.solver == "SRI" else None)
        sol = ODESolution([us[-1]], [t2], nfe)
        return sol, dict(drift=st["drift"], diffusion=st["diffusion"], nfe_drift=nfe, nfe_diffusion=nfe_g,
                         reg_val=reg_val, rng=rng, training=st["training"])

    def pullback(self, x, ps, st, du_end, w_reg=0.0, noise=None):
        """What the reference's Tracker-based pullback gives for  loss = <du_end, sol.u[end]> + w_reg * reg_val
        (test/runtests.jl:361-365, 386-397): (dx, dict(drift=, diffusion=), info).  The forward is re-run with the same
        draws as `__call__` (st['rng']; `noise` if given); the solve is differentiated through its own steps, reg_val
        w.r.t. the parameters only (info['dx_reg'] is None: `gs_x === nothing` in the reference)."""
        if self.solver == "SRI":   # the fixed-grid loop of four-stage SRI steps, newest first (lrnde_sde_sri_step_backward)
            sol, st2 = self(x, ps, st, noise=noise)
            h = self.handle()
            fs = self._last_solve
            n = len(fs["us"])
            ub, dpf, dpg = du_end.contiguous(), None, None
            for i in range(n - 1, -1, -1):
                ui = x if i == 0 else fs["us"][i - 1]
                r = h.sri_step_backward(self.tableau, ui, fs["dW"][i].contiguous(), fs["dZ"][i].contiguous(),
                                        np.float32(fs["t0"] + np.float32(i) * fs["dt"]), fs["dt"], fs["abstol"], fs["reltol"], self.delta,
                                        du_new=ub, dp_drift=dpf, dp_diff=dpg)
                ub, dpf, dpg = r["dx"], r["dp_drift"], r["dp_diff"]
            mode = self.regularize if st["training"] else "none"
            if mode != "none" and w_reg != 0.0:
                lo = self._last_local
                rg = h.sri_step_backward(self.tableau, lo["u1"], lo["dW"], lo["dZ"], lo["t1"], lo["dt"], lo["abstol"], lo["reltol"], self.delta,
                                         du_new=None, w_reg=w_reg, want_dx=False, dp_drift=dpf, dp_diff=dpg)
                assert rg["reg_val"] == st2["reg_val"]
            return ub, dict(drift=dpf, diffusion=dpg), dict(sol=sol, st=st2, dx_reg=None)
        if self.adaptive:
            return self.pullback_series(x, ps, st, None, du_end=du_end, w_reg=w_reg, path=noise)
        sol, st2 = self(x, ps, st, noise=noise)
        h = self.handle()
        fs = self._last_solve
        bw = h.solve_fixed_backward(x, fs["u_traj"], fs["dW"].contiguous(), fs["t0"], fs["dt"], du_end, solver=self.solver)
        dpf, dpg = bw["dp_drift"], bw["dp_diff"]
        mode = self.regularize if st["training"] else "none"
        if mode != "none" and w_reg != 0.0:
            lo = self._last_local
            if self.solver == "RKMil":
                rg = h.rkmil_reg_grad(lo["u1"], lo["dW"], lo["t1"], lo["dt"], lo["abstol"], lo["reltol"])
            else:
                rg = h.euler_heun_reg_grad(lo["u1"], lo["dW"], lo["t1"], lo["dt"], lo["abstol"], lo["reltol"], self.delta)
            assert rg["reg_val"] == st2["reg_val"]
            dpf = dpf + np.float32(w_reg) * rg["dp_drift"]
            dpg = dpg + np.float32(w_reg) * rg["dp_diff"]
        return bw["dx"], dict(drift=dpf, diffusion=dpg), dict(sol=sol, st=st2, dx_reg=None)

    def pullback_series(self, x, ps, st, du_series, du_end=None, w_reg=0.0, path=None, z_local=None):
        """adaptive layer: pullback of  sum_j <du_series[j], sol.u[j]> + w_reg * reg_val  (du_end alone = a cotangent on
        sol.u[end], what `diffeqsol_to_array` passes back).  The forward is re-run with the same draws; the backward is the
        reverse sweep over its recorded accepted steps (lrnde_sde_node_backward_recorded)."""
        sol, st2 = self._call_adaptive(x, ps, st, path=path, z_local=z_local)
        ns = self._last_adaptive["nseries"]
        if du_series is None:
            du_series = torch.zeros((ns,) + tuple(x.shape), dtype=torch.float32, device=x.device)
            du_series[ns - 1] = du_end
        bw = self.handle().node_backward_recorded(du_series, w_reg=w_reg)
        return bw["dx"], dict(drift=bw["dp_drift"], diffusion=bw["dp_diff"]), dict(sol=sol, st=st2, dx_reg=None)
