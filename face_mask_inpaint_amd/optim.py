"""Multi-tensor Adam on the HIP kernel ``fmi_adam_step_f32``: one launch per optimiser step (per 64 tensors)
instead of torch.optim.Adam's per-tensor / foreach kernels.  Same defaults and update arithmetic as
``torch.optim.Adam`` (train_reference_fill.py:309-315: lr only)."""
from __future__ import annotations

import torch

from . import functional as FF


class FusedAdam(torch.optim.Optimizer):
    """``capturable=True``: the step count lives in a device tensor (one per parameter group) and the bias corrections are computed
    on the device, so ``step()`` can be captured in a HIP graph (torch.cuda.graph) together with the forward and backward pass;
    every parameter of a group must then receive a gradient in every step (they share the counter)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable

    @torch.no_grad()
    def _step_capturable(self):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if "step_dev" not in group:
                group["step_dev"] = torch.zeros(1, device=ps[0].device, dtype=torch.int32)
            for p in ps:
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            FF.adam_step([p.data for p in ps], [p.grad.contiguous() for p in ps], [self.state[p]["exp_avg"] for p in ps],
                         [self.state[p]["exp_avg_sq"] for p in ps], group["step_dev"], group["lr"], group["betas"][0], group["betas"][1],
                         group["eps"], group["weight_decay"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            self._step_capturable()
            return loss
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                step = st["step"] if step is None else step
                if st["step"] != step:  # tensors that joined later need their own bias correction
                    FF.adam_step([p.data], [p.grad.contiguous()], [st["exp_avg"]], [st["exp_avg_sq"]], st["step"], group["lr"],
                                 group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"])
                    continue
                ps.append(p.data)
                gs.append(p.grad.contiguous())
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            if ps:
                FF.adam_step(ps, gs, ms, vs, step, group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"])
        return loss
