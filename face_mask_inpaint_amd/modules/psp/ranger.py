"""Host-side mirror of modules/psp/ranger.py: Ranger = RAdam + Lookahead + gradient centralisation (the ``--optimizer ranger``
of train_psp.py:290-293), same constructor and defaults, same per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``,
``slow_buffer``), same update arithmetic and order -- as ONE multi-tensor launch pair per 32 tensors
(``fmi_ranger_step_f32``) instead of a Python loop of ~12 small kernels per parameter."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from ... import _lib
from ... import functional as FF


class Ranger(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, weight_decay=0, use_gc=True,
                 gc_conv_only=False):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha = alpha
        self.k = k
        self.use_gc = use_gc
        self.gc_gradient_threshold = 3 if gc_conv_only else 1

    def _rectification(self, step, beta1, beta2):
        """ranger.py:145-162 (depends on the step count only)"""
        beta2_t = beta2 ** step
        n_sma_max = 2 / (1 - beta2) - 1
        n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma > self.N_sma_threshhold:
            return True, math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** step)
        return False, 1.0 / (1 - beta1 ** step)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Ranger optimizer does not support sparse gradients")
                FF._chk(p.data, p.grad)
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p.data)
                    state["exp_avg_sq"] = torch.zeros_like(p.data)
                    state["slow_buffer"] = p.data.clone()
                state["step"] += 1
                by_step.setdefault(state["step"], []).append(p)
            for step, ps in by_step.items():
                rectified, step_size = self._rectification(step, beta1, beta2)
                rows_total = sum(p.shape[0] for p in ps if self.use_gc and p.grad.dim() > self.gc_gradient_threshold)
                scratch = torch.empty(max(rows_total, 1), device=ps[0].device, dtype=torch.float32)
                entries = (_lib.RangerEntry * len(ps))()
                off = 0
                keep = []
                for e, p in zip(entries, ps):
                    st = self.state[p]
                    g = p.grad.contiguous()
                    keep.append(g)
                    e.p, e.g, e.m, e.v, e.slow = p.data.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["slow_buffer"].data_ptr()
                    e.n = p.numel()
                    if self.use_gc and g.dim() > self.gc_gradient_threshold:  # centralise conv / fc gradients (ranger.py:132-133)
                        e.cols = p.numel() // p.shape[0]
                        e.row_mean = scratch.data_ptr() + 4 * off
                        off += p.shape[0]
                    else:
                        e.cols = p.numel()
                        e.row_mean = None
                _lib.lib().ranger_step_f32(entries, len(ps), group["lr"], beta1, beta2, group["eps"], group["weight_decay"], step_size,
                                           1 if rectified else 0, self.alpha, 1 if step % group["k"] == 0 else 0, FF._st())
        return loss
