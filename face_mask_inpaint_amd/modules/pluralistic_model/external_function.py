"""Host-side mirror of modules/pluralistic_model/external_function.py for the hot path: SpectralNorm,
GANLoss, GramMatrix / StyleLoss, contextual_loss -- same names, arguments and state_dict keys, computed by the
HIP kernels (face_mask_inpaint_amd.functional).  Dead code of the reference (cal_gradient_penalty,
get_features, ContentLoss, img_crop, Normalization) is intentionally absent.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import Parameter

from ... import functional as FF
from ...weights import packed, weight_scope


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    """Reference: external_function.py:16-72.  Holds ``module.weight_bar / weight_u / weight_v`` exactly like the
    reference (so checkpoints load), but the power iteration, sigma and W/sigma are produced for the whole network
    by one ``fmi_weight_prepare_f32`` launch (face_mask_inpaint_amd.weights.weight_scope) instead of per conv.
    ``forward`` then runs the wrapped conv through the implicit-GEMM kernels on NHWC tensors."""

    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        if not 1 <= int(power_iterations) <= 64:
            raise ValueError("power_iterations must be in 1 .. 64")
        if name != "weight" or not isinstance(module, (nn.Conv2d, nn.ConvTranspose2d)):
            raise NotImplementedError("SpectralNorm wraps the 'weight' of Conv2d / ConvTranspose2d only")
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        object.__setattr__(module, "_fmi_power_iterations", int(power_iterations))  # read by weights.weight_scope
        if not hasattr(module, name + "_bar"):
            w = getattr(module, name)
            height = w.data.shape[0]
            width = w.view(height, -1).data.shape[1]
            u = Parameter(l2normalize(w.data.new(height).normal_(0, 1)), requires_grad=False)
            v = Parameter(l2normalize(w.data.new(width).normal_(0, 1)), requires_grad=False)
            w_bar = Parameter(w.data)
            del module._parameters[name]
            module.register_parameter(name + "_u", u)
            module.register_parameter(name + "_v", v)
            module.register_parameter(name + "_bar", w_bar)

    def forward(self, x):
        """NCHW-shaped in / out, like the reference."""
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))

    def nhwc(self, x_nhwc, residual=None, act=FF.ACT_NONE, pad_mode=0, pad=None):
        with weight_scope(self):
            return run_conv(self.module, x_nhwc, residual=residual, act=act, pad_mode=pad_mode, pad=pad)


def run_conv(conv: nn.Module, x_nhwc, residual=None, act=FF.ACT_NONE, pad_mode=0, pad=None, in_act=None, skip_act_bwd=False, passthrough=False):
    """Run a (prepared) nn.Conv2d / nn.ConvTranspose2d parameter holder on an NHWC tensor.  ``in_act`` = ("apply", slope): the
    convolution reads lrelu(x, slope); ("mask", slope): x already is that activation's output -- see functional._Conv2d.
    ``passthrough`` (Conv2d only): returns (y, x') with x' = the input for its OTHER consumer, whose gradient then joins inside this
    convolution's adjoint kernel instead of in a separate accumulation pass."""
    pw = packed(conv)
    if isinstance(conv, nn.ConvTranspose2d):
        if conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] or conv.groups != 1:
            raise NotImplementedError("ConvTranspose2d geometry outside the hot path")
        if in_act is not None:
            if in_act[0] != "apply":
                raise NotImplementedError("ConvTranspose2d with a masked input gradient")
            x_nhwc = FF.leaky_relu(x_nhwc, in_act[1])
        return FF.conv_transpose2d(x_nhwc, pw, conv.bias, residual, conv.stride[0], conv.padding[0], conv.output_padding[0])
    if conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] or conv.groups != 1 or conv.dilation[0] != conv.dilation[1]:
        raise NotImplementedError("Conv2d geometry outside the hot path")
    p = conv.padding[0] if pad is None else pad
    return FF.conv2d(x_nhwc, pw, conv.bias, residual, conv.stride[0], p, pad_mode, act, in_act, skip_act_bwd, conv.dilation[0], passthrough)


class GANLoss(nn.Module):
    """Reference: external_function.py:80-131 (lsgan only on the hot path: loss.py:73)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode != "lsgan":
            raise NotImplementedError("gan mode %s not implemented (the reference trainer uses lsgan)" % gan_mode)
        self._labels = (float(target_real_label), float(target_fake_label))

    def __call__(self, prediction, target_is_real, is_disc=False):
        return FF.mse_to_const(prediction, self._labels[0] if target_is_real else self._labels[1])


def _npc(x):
    """accept NCHW-shaped (any strides) or NHWC-contiguous 4-D features -> [N, P, C] contiguous"""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(n, h * w, c)


def GramMatrix(input):
    """[N,C,H,W] -> [N,C,C] / (C*H*W)  (external_function.py:180-185)"""
    return FF.gram_matrix(_npc(input))


def StyleLoss(input, target):
    target = GramMatrix(target).detach()
    return FF.l1_loss(GramMatrix(input), target)


def contextual_loss(x, y, h=0.5):
    """external_function.py:231-274; x, y are NCHW-shaped features, y carries no gradient."""
    assert x.size() == y.size()
    return FF.contextual_loss(_npc(x), _npc(y).detach(), h)
