"""Host-side mirror of modules/pluralistic_model/base_function.py (hot-path subset): same class names,
constructor arguments, sub-module names and therefore ``state_dict`` keys (``conv1.module.weight_bar``,
``model.N.module.*`` aliases, ``shortcut.0.module.*`` ...), with every forward running on the HIP kernels.

Public ``forward`` methods take / return NCHW-shaped tensors like the reference; internally activations are NHWC
contiguous (``nhwc`` methods), so chained blocks never transpose.  CoordConv / AddCoords, get_scheduler,
print_network and the DataParallel branch of init_net are not on the hot path and are absent.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
from torch.nn import init

from ... import functional as FF
from ...weights import packed, weight_scope
from .external_function import SpectralNorm, run_conv


# ----------------------------------------------------------------------------------------------
# init / factories (base_function.py:14-146)
# ----------------------------------------------------------------------------------------------
def init_weights(net, init_type="normal", gain=0.02):
    """Same rule as the reference: only modules that (still) own a ``weight`` and whose class name contains
    Conv / Linear are re-initialised -- SpectralNorm-wrapped convs lost ``weight`` and keep the default init."""

    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, 0.02)
            init.constant_(m.bias.data, 0.0)

    net.apply(init_func)


def get_norm_layer(norm_type="batch"):
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=True)
    if norm_type == "none":
        return None
    if norm_type == "batch":
        raise NotImplementedError("BatchNorm2d is not on the PICNet hot path (encoder norm=none, decoder norm=instance)")
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def get_nonlinearity_layer(activation_type="PReLU"):
    if activation_type == "ReLU":
        return nn.ReLU()
    if activation_type == "LeakyReLU":
        return nn.LeakyReLU(0.1)
    raise NotImplementedError("activation layer [%s] is not on the hot path" % activation_type)


def init_net(net, init_type="normal", activation="relu", gpu_ids=[]):
    if len(gpu_ids) > 0:
        raise NotImplementedError("multi-GPU runs use one process per GPU (face_mask_inpaint_amd.distributed), not nn.DataParallel")
    init_weights(net, init_type)
    return net


def _freeze(*args):
    for module in args:
        if module:
            for p in module.parameters():
                p.requires_grad = False


def _unfreeze(*args):
    for module in args:
        if module:
            for p in module.parameters():
                p.requires_grad = True


def spectral_norm(module, use_spect=True):
    return SpectralNorm(module) if use_spect else module


def coord_conv(input_nc, output_nc, use_spect=False, use_coord=False, with_r=False, **kwargs):
    if use_coord:
        raise NotImplementedError("CoordConv is dead code in the reference's hot path")
    return spectral_norm(nn.Conv2d(input_nc, output_nc, **kwargs), use_spect)


def _slope(nonlinearity) -> float:
    if isinstance(nonlinearity, nn.LeakyReLU):
        return float(nonlinearity.negative_slope)
    if isinstance(nonlinearity, nn.ReLU):
        return 0.0
    raise NotImplementedError(type(nonlinearity))


def _conv(m):
    """the nn.Conv2d / nn.ConvTranspose2d parameter holder behind an (optionally) spectral-normed layer"""
    return m.module if isinstance(m, SpectralNorm) else m


def _norm_act(norm: nn.Module, x, slope, passthrough=False):
    if not isinstance(norm, nn.InstanceNorm2d) or not norm.affine or norm.track_running_stats:
        raise NotImplementedError("only InstanceNorm2d(affine=True) is on the hot path")
    return FF.instance_norm_act(x, norm.weight, norm.bias, norm.eps, slope, passthrough)


class _NhwcBlock(nn.Module):
    """public forward = NCHW-shaped tensors, like the reference; ``nhwc`` = channels-last fast path"""

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))


# ----------------------------------------------------------------------------------------------
# blocks (base_function.py:207-398)
# ----------------------------------------------------------------------------------------------
class ResBlock(_NhwcBlock):
    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU(),
                 sample_type="none", use_spect=False, use_coord=False):
        super().__init__()
        hidden_nc = output_nc if hidden_nc is None else hidden_nc
        self.sample = sample_type != "none"
        if sample_type == "down":
            self.pool = nn.AvgPool2d(kernel_size=2, stride=2)
        elif sample_type != "none":
            raise NotImplementedError("sample type [%s] is not on the hot path" % sample_type)
        self.conv1 = coord_conv(input_nc, hidden_nc, use_spect, use_coord, kernel_size=3, stride=1, padding=1)
        self.conv2 = coord_conv(hidden_nc, output_nc, use_spect, use_coord, kernel_size=3, stride=1, padding=1)
        self.bypass = coord_conv(input_nc, output_nc, use_spect, use_coord, kernel_size=1, stride=1, padding=0)
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, self.conv1, nonlinearity, self.conv2)
            self._norms = None
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, self.conv1, norm_layer(hidden_nc), nonlinearity, self.conv2)
            self._norms = (0, 3)
        self.shortcut = nn.Sequential(self.bypass)
        self._slope = _slope(nonlinearity)

    def nhwc(self, x):
        with weight_scope(self):
            # x has two consumers (the main path and the 1x1 bypass): the bypass reads the pass-through copy the main path's first op
            # hands on, so its gradient re-enters that op's backward kernel instead of an accumulation pass of its own
            if self._norms is None:
                # LeakyReLU -> conv pairs: the activation is applied on the way in and differentiated in the adjoint's epilogue
                act_in = ("apply", self._slope)
                h, xp = run_conv(_conv(self.conv1), x, in_act=act_in, passthrough=True)
                s = run_conv(_conv(self.bypass), xp)
                out = run_conv(_conv(self.conv2), h, residual=s, in_act=act_in)  # model(x) + shortcut(x), fused in the epilogue
            else:
                h, xp = _norm_act(self.model[0], x, self._slope, passthrough=True)
                s = run_conv(_conv(self.bypass), xp)
                h = run_conv(_conv(self.conv1), h)
                h = _norm_act(self.model[3], h, self._slope)
                out = run_conv(_conv(self.conv2), h, residual=s)  # model(x) + shortcut(x), fused in the epilogue
            if self.sample:  # pool(a) + pool(b) == pool(a + b)
                out = FF.avg_pool(out, 2)
            return out


class ResBlockEncoderOptimized(_NhwcBlock):
    def __init__(self, input_nc, output_nc, norm_layer=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU(), use_spect=False, use_coord=False):
        super().__init__()
        if norm_layer is not None:
            raise NotImplementedError("ResBlockEncoderOptimized runs with norm='none' on the hot path")
        self.conv1 = coord_conv(input_nc, output_nc, use_spect, use_coord, kernel_size=3, stride=1, padding=1)
        self.conv2 = coord_conv(output_nc, output_nc, use_spect, use_coord, kernel_size=3, stride=1, padding=1)
        self.bypass = coord_conv(input_nc, output_nc, use_spect, use_coord, kernel_size=1, stride=1, padding=0)
        self.model = nn.Sequential(self.conv1, nonlinearity, self.conv2, nn.AvgPool2d(kernel_size=2, stride=2))
        self.shortcut = nn.Sequential(nn.AvgPool2d(kernel_size=2, stride=2), self.bypass)
        self._slope = _slope(nonlinearity)

    def nhwc(self, x):
        with weight_scope(self):
            h, xp = run_conv(_conv(self.conv1), x, passthrough=True)
            s = run_conv(_conv(self.bypass), FF.avg_pool(xp, 2))
            h = run_conv(_conv(self.conv2), h, in_act=("apply", self._slope))
            return FF.add(FF.avg_pool(h, 2), s)


class ResBlockDecoder(_NhwcBlock):
    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU(),
                 use_spect=False, use_coord=False):
        super().__init__()
        hidden_nc = output_nc if hidden_nc is None else hidden_nc
        self.conv1 = spectral_norm(nn.Conv2d(input_nc, hidden_nc, kernel_size=3, stride=1, padding=1), use_spect)
        self.conv2 = spectral_norm(nn.ConvTranspose2d(hidden_nc, output_nc, kernel_size=3, stride=2, padding=1, output_padding=1), use_spect)
        self.bypass = spectral_norm(nn.ConvTranspose2d(input_nc, output_nc, kernel_size=3, stride=2, padding=1, output_padding=1), use_spect)
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, self.conv1, nonlinearity, self.conv2)
            self._norms = None
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, self.conv1, norm_layer(hidden_nc), nonlinearity, self.conv2)
            self._norms = (0, 3)
        self.shortcut = nn.Sequential(self.bypass)
        self._slope = _slope(nonlinearity)
        self._pair_geometry = True  # both ConvTranspose2d are kernel 3 / stride 2 / padding 1 / output_padding 1 by construction

    def nhwc(self, x):
        with weight_scope(self):
            c2, cb = _conv(self.conv2), _conv(self.bypass)
            if self._norms is None:
                h, xp = run_conv(_conv(self.conv1), x, in_act=("apply", self._slope), passthrough=True)
                h = FF.leaky_relu(h, self._slope)
            else:
                h, xp = _norm_act(self.model[0], x, self._slope, passthrough=True)  # the bypass ConvTranspose2d's gradient joins in the IN backward
                h = run_conv(_conv(self.conv1), h)
                h = _norm_act(self.model[3], h, self._slope)
            # main path + bypass: thin outputs on a large map go out as ONE launch (the reduction continues over the bypass input's
            # channels: the intermediate is neither written nor re-read), otherwise the bypass result is the main convolution's residual
            pw2, pwb = packed(c2), packed(cb)
            if self._pair_geometry and FF.conv_transpose2d_pair_ok(h, pw2, xp, pwb):
                bias = c2.bias if cb.bias is None else (cb.bias if c2.bias is None else FF.add(c2.bias, cb.bias))
                return FF.conv_transpose2d_pair(h, pw2, xp, pwb, bias)
            s = run_conv(cb, xp)
            return run_conv(c2, h, residual=s)


class Output(_NhwcBlock):
    def __init__(self, input_nc, output_nc, kernel_size=3, norm_layer=nn.BatchNorm2d, nonlinearity=nn.LeakyReLU(),
                 use_spect=False, use_coord=False):
        super().__init__()
        if norm_layer is not None:
            raise NotImplementedError("Output runs without a norm layer on the hot path (network.py:233-236)")
        self.conv1 = coord_conv(input_nc, output_nc, use_spect, use_coord, kernel_size=kernel_size, padding=0, bias=True)
        self._pad = int(kernel_size / 2)
        self.model = nn.Sequential(nonlinearity, nn.ReflectionPad2d(self._pad), self.conv1, nn.Tanh())
        self._slope = _slope(nonlinearity)

    def nhwc(self, x):
        with weight_scope(self):
            conv = _conv(self.conv1)
            if self._pad == 1 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.groups == 1 and conv.dilation == (1, 1):
                # LeakyReLU, ReflectionPad2d, the convolution and tanh in one kernel (thin-output path); falls back to
                # LeakyReLU + convolution by itself for shapes that path does not take
                return FF.lrelu_conv2d(x, packed(conv), conv.bias, self._slope, 1, 1, FF.ACT_TANH)
            h = FF.leaky_relu(x, self._slope)
            # ReflectionPad2d is folded into the gather of the implicit GEMM, tanh into its epilogue
            return run_conv(conv, h, act=FF.ACT_TANH, pad_mode=1, pad=self._pad)


class Auto_Attn(nn.Module):
    """Short+Long attention (base_function.py:401-448).  Only the ``pre is None`` form is on the hot path; the
    attention map itself is never materialised, so the second return value is None (every caller discards it:
    network.py:265-268, 364-366)."""

    def __init__(self, input_nc, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.input_nc = input_nc
        self.query_conv = nn.Conv2d(input_nc, input_nc // 4, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.alpha = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)
        self.model = ResBlock(int(input_nc * 2), input_nc, input_nc, norm_layer=norm_layer, use_spect=True)
        self.model._fmi_never_runs = True  # the reference never executes it when pre is None; its u/v must not advance

    def nhwc(self, x):
        with weight_scope(self):
            n, h, w, c = x.shape
            q = run_conv(self.query_conv, x)
            (o,) = FF.self_attention(q.view(n, h * w, -1), [x.view(n, h * w, c)])
            return FF.scale_add_param(o.view(n, h, w, c), self.gamma, x)

    def forward(self, x, pre=None, mask=None):
        if pre is not None:
            raise NotImplementedError("the long-term (pre/mask) branch of Auto_Attn is never taken by the reference's callers")
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x))), None
