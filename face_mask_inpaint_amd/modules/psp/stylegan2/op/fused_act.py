"""Host-side mirror of modules/psp/stylegan2/op/fused_act.py: FusedLeakyReLU / fused_leaky_relu with first- and
second-order autograd on ``fmi_fused_bias_act_f32`` (contiguous NC... layout, bias index = dim 1)."""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from ..... import _lib
from .....functional import _chk, _p, _st


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """the reference's pybind entry (op/fused_bias_act.cpp:11-20); empty bias / refer tensors mean 'absent'"""
    x = input.contiguous()
    b = bias.contiguous() if bias is not None and bias.numel() else None
    r = refer.contiguous() if refer is not None and refer.numel() else None
    bf16 = x.dtype == torch.bfloat16
    if bf16:
        b = b.to(torch.bfloat16) if b is not None else None
        r = r.to(torch.bfloat16) if r is not None else None
    _chk(x, b, r, dtype=x.dtype if bf16 else torch.float32)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    y = torch.empty_like(x)
    if x.numel():
        fn = _lib.lib().fused_bias_act_bf16 if bf16 else _lib.lib().fused_bias_act_f32
        fn(_p(x), _p(b), _p(r), _p(y), x.numel(), step_b, b.numel() if b is not None else 1, act, grad, float(alpha), float(scale), _st())
    return y


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = fused_bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
        n, c = grad_input.shape[0], grad_input.shape[1]
        hw = grad_input.numel() // (n * c)
        grad_bias = torch.zeros(c, device=grad_input.device, dtype=torch.float32)
        # the sum reads the cotangent in ITS dtype (a bf16 buffer handed to the fp32 entry would be read out of bounds)
        fn = _lib.lib().bias_grad_nchw_bf16 if grad_input.dtype == torch.bfloat16 else _lib.lib().bias_grad_nchw_f32
        _chk(grad_input, dtype=grad_input.dtype if grad_input.dtype == torch.bfloat16 else torch.float32)
        if grad_input.numel():
            fn(_p(grad_input), n, c, hw, _p(grad_bias), _st())
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        gradgrad_out = fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
