"""Host-side mirror of modules/psp/stylegan2/op/upfirdn2d.py: ``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))``
with first- and second-order autograd, on the HIP kernels ``fmi_upfirdn2d_f32`` / ``fmi_upfirdn2d_bf16`` (no JIT build at
import, any up/down/kernel size; bf16 tensors are filtered in fp32 and rounded to nearest even).  ``upfirdn2d_native`` of the reference is test infrastructure there and lives in oracle/ here."""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function

from ..... import _lib
from .....functional import FmiError, _chk, _p, _st


def _native(x_planes, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """the reference's pybind entry (op/upfirdn2d.cpp:12-23) for minor = 1: [major,H,W] -> [major,OH,OW]"""
    bf16 = x_planes.dtype == torch.bfloat16  # the reference's op is dispatched on the tensor dtype, taps included
    if bf16:
        kernel = kernel.to(torch.bfloat16)
    _chk(x_planes, kernel, dtype=x_planes.dtype if bf16 else torch.float32)
    major, in_h, in_w = x_planes.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    if out_h <= 0 or out_w <= 0:
        raise FmiError("upfirdn2d: empty output")
    out = torch.empty((major, out_h, out_w), device=x_planes.device, dtype=x_planes.dtype)
    fn = _lib.lib().upfirdn2d_bf16 if bf16 else _lib.lib().upfirdn2d_f32
    fn(_p(x_planes), _p(kernel), _p(out), major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, _st())
    return out


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        go = grad_output.contiguous().reshape(-1, out_size[0], out_size[1])
        gi = _native(go, grad_kernel, down_x, down_y, up_x, up_y, g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return gi.view(in_size[0], in_size[1], in_size[2], in_size[3])

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        ggi = gradgrad_input.contiguous().reshape(-1, in_size[2], in_size[3])
        ggo = _native(ggi, kernel, up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3])
        return ggo.view(in_size[0], in_size[1], out_size[0], out_size[1]), None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        x = input.contiguous().reshape(-1, in_h, in_w)
        kernel = kernel.contiguous()
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        out = _native(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
        out_h, out_w = out.shape[1], out.shape[2]
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        ctx.g_pad = (kernel_w - pad_x0 - 1, in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kernel_h - pad_y0 - 1, in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        gi = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad, ctx.in_size, ctx.out_size)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
