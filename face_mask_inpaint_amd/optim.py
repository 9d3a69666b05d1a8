"""Multi-tensor Adam on the HIP kernel ``fmi_adam_step_f32``: one launch per optimiser step (per 64 tensors)
instead of torch.optim.Adam's per-tensor / foreach kernels.  Same defaults and update arithmetic as
``torch.optim.Adam`` (train_reference_fill.py:309-315: lr only)."""
from __future__ import annotations

import torch

from . import functional as FF


class FusedAdam(torch.optim.Optimizer):
    """``capturable=True``: the step count lives in a device tensor (one per parameter group) and the bias corrections are computed
    on the device, so ``step()`` can be captured in a HIP graph (torch.cuda.graph) together with the forward and backward pass;
    every parameter of a group must then receive a gradient in every step (they share the counter).

    State layout (both modes): ``state[p] = {"step", "exp_avg", "exp_avg_sq"}`` like torch.optim.Adam -- in capturable mode ``step`` is
    the group's shared 1-element int32 device counter, otherwise a python int -- so a state_dict written in one mode loads in the other
    (and after ``load_state_dict`` the counter is rebuilt on the parameters' device).

    ``step(guard=loss)`` (capturable mode): a device scalar; when it is not finite the whole step is a no-op on the device
    (train_psp.py:328-331 skips non-finite losses with a host read, which a captured graph cannot do).

    A captured graph bakes ``lr`` / ``betas`` / ``eps`` / ``weight_decay`` into its launches: changing ``param_groups[i]["lr"]`` needs a
    re-capture."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable

    @staticmethod
    def _steps_taken(st) -> int:
        v = st.get("step", 0)
        return int(v.item()) if torch.is_tensor(v) else int(v)

    def _counter(self, group, ps):
        """the group's device counter, (re)built on the parameters' device from whatever the state holds"""
        dev = ps[0].device
        c = group.get("step_dev")
        if c is None or c.device != dev:
            taken = max([self._steps_taken(self.state[p]) for p in ps if self.state[p]] + [int(c.item()) if c is not None else 0])
            c = group["step_dev"] = torch.full((1,), taken, device=dev, dtype=torch.int32)
        return c

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group.pop("step_dev", None)  # rebuilt from state[p]["step"] on first use, on the right device

    @torch.no_grad()
    def _step_capturable(self, guard=None):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            counter = self._counter(group, ps)
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = counter
            FF.adam_step([p.data for p in ps], [p.grad.contiguous() for p in ps], [self.state[p]["exp_avg"] for p in ps],
                         [self.state[p]["exp_avg_sq"] for p in ps], counter, group["lr"], group["betas"][0], group["betas"][1],
                         group["eps"], group["weight_decay"], guard=guard)

    @torch.no_grad()
    def step(self, closure=None, guard=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            self._step_capturable(guard)
            return loss
        if guard is not None:
            raise FF.FmiError("step(guard=...) needs FusedAdam(capturable=True)")
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = self._steps_taken(st) + 1
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
