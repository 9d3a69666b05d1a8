"""Multi-tensor Adam on the HIP kernel ``fmi_adam_step_f32``: one launch per optimiser step (per 64 tensors)
instead of torch.optim.Adam's per-tensor / foreach kernels.  Same defaults and update arithmetic as
``torch.optim.Adam`` (train_reference_fill.py:309-315: lr only)."""
from __future__ import annotations

import torch

from . import functional as FF


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
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
