"""Optimisers and learning-rate schedulers of the experiments: host mirror of
experiments/src/construct.jl:104-152 (`construct(expt, cfg::OptimizerConfig)`) and experiments/src/utils.jl:1-68.
The schedulers are host scalars (same formulas, 1-based step `t` as the training loops call them); the update rules
run on the device in one fused pass (lrnde_opt_update).  Optimisers.jl is un-vendored: its rules are restated from
their documented formulas (UPSTREAM-RECALL)."""
import bisect
import ctypes as C
import math

import torch

from . import _lib as L


# ----- schedulers: experiments/src/utils.jl:1-68 -----
class ExponentialDecay:
    """:2-13: lr0 * exp(-k t), k = log(lr0/lr1)/nsteps"""

    def __init__(self, lr0, lr1, nsteps):
        self.lr0, self.lr1, self.nsteps = float(lr0), float(lr1), int(nsteps)
        self.k = math.log(self.lr0 / self.lr1) / self.nsteps

    def __call__(self, t):
        return self.lr0 * math.exp(-self.k * t)


class InverseDecay:
    """:15-24: lr0 / (1 + gamma t)"""

    def __init__(self, lr0, gamma):
        self.lr0, self.gamma = float(lr0), float(gamma)

    def __call__(self, t):
        return self.lr0 / (1 + self.gamma * t)


class Step:
    """:26-38: lr0 * gamma^(searchsortedfirst(step_sizes, t - 1) - 1)"""

    def __init__(self, lr0, gamma, step_sizes):
        self.lr0, self.gamma = float(lr0), float(gamma)
        self.step_sizes = [step_sizes] if isinstance(step_sizes, int) else list(step_sizes)

    def __call__(self, t):
        return self.lr0 * self.gamma ** bisect.bisect_left(self.step_sizes, t - 1)  # searchsortedfirst - 1 (0-based insertion point)


class Constant:
    def __init__(self, lr):
        self.lr = float(lr)

    def __call__(self, t):
        return self.lr


class CosineAnneal:
    """:46-68; restart=True: warm restarts every `period` steps, each cycle divided by dampen^cycle"""

    def __init__(self, lr0, lr1, period, restart=False, dampen=1.0):
        self.range, self.offset = abs(lr0 - lr1), min(lr0, lr1)
        self.dampen, self.period, self.restart = float(dampen), int(period), bool(restart)

    def __call__(self, t):
        if self.restart:
            d = self.dampen ** ((t - 1) // self.period)
            return (self.range * (1 + math.cos(math.pi * ((t - 1) % self.period) / self.period)) / 2 + self.offset) / d
        return self.range * (1 + math.cos(math.pi * (t - 1) / self.period)) / 2 + self.offset


def construct_scheduler(name, learning_rate, total_steps=None, **cfg):
    """construct.jl:128-150"""
    if name == "cosine":
        return CosineAnneal(learning_rate, learning_rate / cfg["cosine_lr_div_factor"], cfg["cosine_cycle_length"], restart=True,
                            dampen=cfg.get("cosine_dampen", 1.0))
    if name == "constant":
        return Constant(learning_rate)
    if name == "step":
        return Step(learning_rate, cfg["step_lr_step_decay"], cfg["step_lr_steps"])
    if name == "inverse":
        return InverseDecay(learning_rate, cfg["inverse_decay_factor"])
    if name == "exponential":
        return ExponentialDecay(learning_rate, learning_rate / cfg["exponential_lr_div_factor"], total_steps)
    raise ValueError(f"unknown value for `scheduler` = {name}. Supported options are: `constant`, `step`, `exponential`, "
                     "`inverse` and `cosine`.")


# ----- update rules: construct.jl:104-126 -----
_KIND = {"sgd": 0, "descent": 0, "momentum": 1, "nesterov": 2, "adam": 3, "adamw": 3, "adamax": 4}


class Optimiser:
    """One rule over any number of flat float32 CUDA parameter vectors; `update(params, grads, lr=None)` in place.
    `optimizer` as in the YAML configs ("adam", "adamw", "adamax", "sgd" with momentum / nesterov)."""

    def __init__(self, optimizer="adam", learning_rate=1e-3, momentum=0.0, nesterov=False, weight_decay=0.0, beta=(0.9, 0.999),
                 eps=1e-8):
        if optimizer not in ("adam", "adamw", "adamax", "sgd"):
            raise ValueError(f"unknown value for `optimizer` = {optimizer}. Supported options are: `adam`, `adamax` and `sgd`.")
        if optimizer == "sgd":
            self.kind = 2 if nesterov else (0 if momentum == 0 else 1)
        else:
            self.kind = _KIND[optimizer]
        self.lr, self.rho, self.beta, self.eps = float(learning_rate), float(momentum), (float(beta[0]), float(beta[1])), float(eps)
        # AdamW(eta) = OptimiserChain(Adam(eta), WeightDecay(0)) upstream; the experiments' own cfg.weight_decay chains WeightDecay
        self.weight_decay = float(weight_decay)
        self.step = 0
        self._state = {}

    def update(self, params, grads, lr=None):
        params = params if isinstance(params, (list, tuple)) else [params]
        grads = grads if isinstance(grads, (list, tuple)) else [grads]
        self.step += 1
        eta = self.lr if lr is None else float(lr)
        for i, (p, g) in enumerate(zip(params, grads)):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_cuda and g.dtype == torch.float32
                    and g.is_contiguous() and g.numel() == p.numel()):
                raise ValueError("parameters and gradients must be contiguous float32 CUDA tensors of equal size")
            if i not in self._state:
                self._state[i] = (torch.zeros_like(p) if self.kind != 0 else None, torch.zeros_like(p) if self.kind >= 3 else None)
            s1, s2 = self._state[i]
            b1 = self.rho if self.kind in (1, 2) else self.beta[0]
            rc = L.lib.lrnde_opt_update(self.kind, C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()),
                                        C.c_void_p(s1.data_ptr()) if s1 is not None else None,
                                        C.c_void_p(s2.data_ptr()) if s2 is not None else None, p.numel(), eta, b1, self.beta[1],
                                        self.eps, self.step, self.weight_decay, p.device.index or 0,
                                        C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
            if rc != 0:
                raise L.LrndeError(rc, "lrnde_opt_update failed")
