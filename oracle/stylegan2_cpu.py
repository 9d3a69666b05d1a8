"""ORACLE (test infrastructure, NOT product code): CPU restatement of the pSp / StyleGAN2 decoder pieces of the hot
path.  The two native ops are plain C (oracle/stylegan2_ops.c); the modulated-conv
blocks are functional torch on a flat parameter dictionary.  Pinned by tests/golden/stylegan2_ops.pt, which was
produced by the reference's own upfirdn2d_native / ModulatedConv2d / StyledConv / ToRGB (oracle/gen_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_sg2.so")
_lib = None


def clib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
    return _lib


def upfirdn2d_planes(x: torch.Tensor, k: torch.Tensor, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1) -> torch.Tensor:
    """x [major, H, W] fp32 -> [major, out_h, out_w]   (op/upfirdn2d.py:150-184)"""
    x = x.contiguous().float()
    k = k.contiguous().float()
    major, h, w = x.shape
    kh, kw = k.shape
    out_h = (h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = torch.empty(major, out_h, out_w)
    rc = clib().oracle_upfirdn2d(C.c_void_p(x.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(out.data_ptr()), major, h, w, kh, kw,
                                 up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    assert rc == 0
    return out


def upfirdn2d(x: torch.Tensor, k: torch.Tensor, up=1, down=1, pad=(0, 0)) -> torch.Tensor:
    """python-level op of the reference (op/upfirdn2d.py:142-147) on [N,C,H,W]"""
    n, c, h, w = x.shape
    o = upfirdn2d_planes(x.reshape(n * c, h, w), k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return o.view(n, c, o.shape[1], o.shape[2])


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    x = x.contiguous().float()
    out = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    bp = C.c_void_p(b.contiguous().data_ptr()) if b is not None and b.numel() else None
    rp = C.c_void_p(ref.contiguous().data_ptr()) if ref is not None and ref.numel() else None
    clib().oracle_fused_bias_act(C.c_void_p(x.data_ptr()), bp, rp, C.c_void_p(out.data_ptr()), C.c_int64(x.numel()), step_b,
                                 b.numel() if bp else 1, act, grad, C.c_float(alpha), C.c_float(scale))
    return out


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


def mask_binarise(m: torch.Tensor) -> torch.Tensor:
    m = m.contiguous()
    out = torch.empty(m.shape, dtype=torch.float32)
    clib().oracle_mask_binarise(C.c_void_p(m.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(m.numel()))
    return out


# ---- differentiable torch restatement of the decoder blocks (stylegan2/model.py) -------------------------------
def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d_t(x, k, up=1, down=1, pad=(0, 0)):
    """differentiable torch form of the same op (zero-insert, pad/crop, conv with flipped kernel, stride)"""
    n, c, h, w = x.shape
    kh, kw = k.shape
    y = x.reshape(n * c, 1, h, 1, w, 1)
    y = F.pad(y, [0, up - 1, 0, 0, 0, up - 1])
    y = y.reshape(n * c, 1, h * up, w * up)
    y = F.pad(y, [max(pad[0], 0), max(pad[1], 0), max(pad[0], 0), max(pad[1], 0)])
    y = y[:, :, max(-pad[0], 0): y.shape[2] - max(-pad[1], 0), max(-pad[0], 0): y.shape[3] - max(-pad[1], 0)]
    y = F.conv2d(y, torch.flip(k, [0, 1]).view(1, 1, kh, kw))
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def equal_linear(P, prefix, x, lr_mul=1.0):
    w = P[prefix + ".weight"]
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    return F.linear(x, w * scale, bias=P[prefix + ".bias"] * lr_mul)


def modulated_conv(P, prefix, x, style, demodulate=True, upsample=False, blur_kernel=(1, 3, 3, 1)):
    """ModulatedConv2d.forward (stylegan2/model.py:241-279), plain and upsample branches"""
    w = P[prefix + ".weight"]  # [1, out, in, k, k]
    _, oc, ic, ks, _ = w.shape
    b = x.shape[0]
    s = equal_linear(P, prefix + ".modulation", style).view(b, 1, ic, 1, 1)
    weight = (1 / math.sqrt(ic * ks * ks)) * w * s
    if demodulate:
        weight = weight * torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8).view(b, oc, 1, 1, 1)
    h, wd = x.shape[2:]
    if upsample:
        wt = weight.transpose(1, 2).reshape(b * ic, oc, ks, ks)
        out = F.conv_transpose2d(x.reshape(1, b * ic, h, wd), wt, padding=0, stride=2, groups=b)
        out = out.view(b, oc, out.shape[2], out.shape[3])
        p = (len(blur_kernel) - 2) - (ks - 1)
        k = make_kernel(list(blur_kernel)) * 4
        return upfirdn2d_t(out, k, pad=((p + 1) // 2 + 1, p // 2 + 1))
    out = F.conv2d(x.reshape(1, b * ic, h, wd), weight.view(b * oc, ic, ks, ks), padding=ks // 2, groups=b)
    return out.view(b, oc, out.shape[2], out.shape[3])


def styled_conv(P, prefix, x, style, noise, upsample=False):
    out = modulated_conv(P, prefix + ".conv", x, style, True, upsample)
    out = out + P[prefix + ".noise.weight"] * noise
    return F.leaky_relu(out + P[prefix + ".activate.bias"].view(1, -1, 1, 1), 0.2) * math.sqrt(2)


def to_rgb(P, prefix, x, style, skip=None):
    out = modulated_conv(P, prefix + ".conv", x, style, demodulate=False) + P[prefix + ".bias"]
    if skip is not None:
        k = make_kernel([1, 3, 3, 1]) * 4
        out = out + upfirdn2d_t(skip, k, up=2, pad=(2, 1))
    return out


def generator_forward(P, latent, noises, size):
    """Generator.forward with input_is_latent=True and explicit per-layer noises (stylegan2/model.py:479-550).
    Composition of the block functions above (which are pinned by the golden fixtures); the whole generator is not
    pinned by a fixture of its own (its 512-channel layers would make it >100 MB)."""
    n = latent.shape[0]
    log_size = int(math.log2(size))
    out = P["input.input"].repeat(n, 1, 1, 1)
    out = styled_conv(P, "conv1", out, latent[:, 0], noises[0])
    skip = to_rgb(P, "to_rgb1", out, latent[:, 1])
    i = 1
    for l in range(log_size - 2):
        out = styled_conv(P, f"convs.{2 * l}", out, latent[:, i], noises[1 + 2 * l], upsample=True)
        out = styled_conv(P, f"convs.{2 * l + 1}", out, latent[:, i + 1], noises[2 + 2 * l])
        skip = to_rgb(P, f"to_rgbs.{l}", out, latent[:, i + 2], skip)
        i += 2
    return skip


def mapping_network(P, z, n_mlp, lr_mul=0.01, prefix="style"):
    """Generator.style: PixelNorm (stylegan2/model.py:10-15) + n_mlp x EqualLinear(activation='fused_lrelu', lr_mul) (:135-172);
    the Sequential index of layer i is i + 1"""
    w = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(n_mlp):
        wt = P[f"{prefix}.{i + 1}.weight"]
        scale = (1 / math.sqrt(wt.shape[1])) * lr_mul
        w = F.leaky_relu(F.linear(w, wt * scale) + P[f"{prefix}.{i + 1}.bias"] * lr_mul, 0.2) * math.sqrt(2)
    return w


def generator_styles_forward(P, styles, noises, size, n_mlp, inject_index=None, truncation=1.0, truncation_latent=None,
                             input_is_latent=False):
    """Generator.forward for a list of style codes (stylegan2/model.py:491-526): mapping network, truncation, repetition of one
    code over all layers or style mixing of two codes at ``inject_index``; returns (image, latent, last feature map)"""
    n_latent = int(math.log2(size)) * 2 - 2
    if not input_is_latent:
        styles = [mapping_network(P, s, n_mlp) for s in styles]
    if truncation < 1:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    else:
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    n = latent.shape[0]
    log_size = int(math.log2(size))
    out = P["input.input"].repeat(n, 1, 1, 1)
    out = styled_conv(P, "conv1", out, latent[:, 0], noises[0])
    skip = to_rgb(P, "to_rgb1", out, latent[:, 1])
    i = 1
    for l in range(log_size - 2):
        out = styled_conv(P, f"convs.{2 * l}", out, latent[:, i], noises[1 + 2 * l], upsample=True)
        out = styled_conv(P, f"convs.{2 * l + 1}", out, latent[:, i + 1], noises[2 + 2 * l])
        skip = to_rgb(P, f"to_rgbs.{l}", out, latent[:, i + 2], skip)
        i += 2
    return skip, latent, out
