"""Host-side mirror of modules/psp/stylegan2/model.py (the classes pSp instantiates: PixelNorm, make_kernel, Upsample,
Blur, EqualLinear, ModulatedConv2d, NoiseInjection, ConstantInput, StyledConv, ToRGB, Generator) with the reference's
constructor / forward signatures and ``state_dict`` keys (``conv.weight`` [1,O,I,k,k], ``conv.modulation.weight/bias``,
``noise.weight``, ``activate.bias``, ``noises.noise_i`` ...).  Downsample / EqualConv2d / ConvLayer / ResBlock /
Discriminator are never instantiated by the reference and are absent.

ModulatedConv2d runs as  demod[n,o] * conv(x * s[n,c], W)  on the shared-weight implicit-GEMM kernels: the same algebra
as the reference's per-sample weights (model.py:244-250) without materialising N x O x I x k x k weights, so all samples
go through ONE GEMM.  Public forwards take / return NCHW-shaped tensors; ``nhwc`` methods chain without transposes.
"""
from __future__ import annotations

import math
import os
import random

import torch
from torch import nn

from .... import functional as FF
from ....weights import packed, weight_scope
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401  (re-exported like the reference)


class PixelNorm(nn.Module):
    def forward(self, input):
        n, c = input.shape
        x = input.contiguous()
        inv = FF.rsqrt_eps(FF._Scale.apply(FF.sqsum_last(x), 1.0 / c), 1e-8)  # rsqrt(mean(x^2) + 1e-8)
        return FF.scale_channels(x.view(n, c, 1), inv.view(n, 1)).view(n, c)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        kernel = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", kernel)
        p = kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def nhwc(self, x):
        return FF.upfirdn2d_nhwc(x, self.kernel, up=self.factor, down=1, pad=self.pad)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        self.separable = torch.as_tensor(kernel).ndim == 1  # make_kernel builds the outer product of a 1-D list
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def nhwc(self, x):
        return FF.upfirdn2d_nhwc(x, self.kernel, pad=self.pad, separable=self.separable)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        b = None
        if self.bias is not None:
            b = self.bias if self.lr_mul == 1 else FF._Scale.apply(self.bias, float(self.lr_mul))
        if self.activation:
            out = FF.linear(input, self.weight, None, self.scale)
            return FF.noise_bias_act(out, b, None, None, 0.2, 2 ** 0.5)
        return FF.linear(input, self.weight, b, self.scale)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError("the downsample branch is never used by the reference's Generator")
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    def nhwc(self, x, style, bias=None, residual=None):
        """x [N,H,W,Cin]; returns [N,H',W',Cout] (+ bias[c] + residual, fused into the GEMM epilogue when no demod)"""
        w = self.weight[0]
        k = self.kernel_size
        s = FF._Scale.apply(self.modulation(style), float(self.scale))  # scale * s[n, i]
        xs = FF.scale_channels(x, s)
        if self.upsample:
            # conv_transpose2d(stride 2, pad 0) = adjoint of the stride-2 conv whose weight is [I][O][k][k]
            (pw,) = FF.prepare_weights([(w.transpose(0, 1).contiguous(), None, None, x.dtype == torch.float32)])
            out = FF.conv_transpose2d(xs, pw, None, None, stride=2, pad=0, out_pad=0)
            out = self.blur.nhwc(out)
        else:
            (pw,) = FF.prepare_weights([(w, None, None, x.dtype == torch.float32)])
            fuse = not self.demodulate
            out = FF.conv2d(xs, pw, bias if fuse else None, residual if fuse else None, 1, self.padding)
            if fuse:
                return out
        if self.demodulate:
            wsq = FF.sqsum_last(w.reshape(self.out_channel * self.in_channel, k * k)).view(self.out_channel, self.in_channel)
            d = FF.rsqrt_eps(FF.linear(FF.mul(s, s), wsq), 1e-8)  # rsqrt(sum_i s^2 sum_k W^2 + 1e-8)  [N, O]
            out = FF.scale_channels(out, d)
        if bias is not None or residual is not None:
            raise NotImplementedError
        return out

    def nhwc_styled(self, x, style, noise, nw, bias, slope, gain):
        """the whole StyledConv on bf16 activations with its output stage fused (model.py:241-294 in two launches + one scaling pass):
        up-convolution -> [Blur + demodulation + noise + bias + leaky ReLU], or [convolution + the same output stage]; None where the
        fused kernels do not take the shape.  noise: [N,1,H',W'] or None (drawn here, NoiseInjection's default)"""
        if x.dtype != torch.bfloat16 or not self.demodulate:
            return None
        w = self.weight[0]
        k = self.kernel_size
        n, h, wd, _ = x.shape
        if self.upsample:
            if not (self.blur.separable and self.out_channel % 32 == 0 and tuple(self.blur.kernel.shape) == (4, 4)):
                return None
            (pw,) = FF.prepare_weights([(w.transpose(0, 1).contiguous(), None, None, False)])
            oh, ow = 2 * h, 2 * wd
        else:
            (pw,) = FF.prepare_weights([(w, None, None, False)])
            if not FF.styled_conv_fused_ok(x, pw, self.padding):
                return None
            oh, ow = h, wd
        if noise is None:
            noise = torch.randn((n, oh, ow), device=x.device)
        else:
            noise = noise.expand(n, 1, oh, ow).reshape(n, oh, ow).contiguous()
        s = FF._Scale.apply(self.modulation(style), float(self.scale))  # scale * s[n, i]
        xs = FF.scale_channels(x, s)
        wsq = FF.sqsum_last(w.reshape(self.out_channel * self.in_channel, k * k)).view(self.out_channel, self.in_channel)
        d = FF.rsqrt_eps(FF.linear(FF.mul(s, s), wsq), 1e-8)  # rsqrt(sum_i s^2 sum_k W^2 + 1e-8)  [N, O]
        if self.upsample:
            u = FF.conv_transpose2d(xs, pw, None, None, stride=2, pad=0, out_pad=0)
            return FF.blur_act(u, self.blur.kernel, self.blur.pad, d, noise, nw, bias, slope, gain)
        return FF.styled_conv(xs, pw, d, noise, nw, bias, slope, gain, self.padding)

    def forward(self, input, style):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(input), style))


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        x = FF.to_nhwc(image)
        n, h, w, c = x.shape
        return FF.to_nchw(FF.noise_bias_act(x, None, noise.expand(n, 1, h, w).reshape(n, h, w).contiguous(), self.weight, 1.0, 1.0))


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


FUSE_STYLED = os.environ.get("FMI_FUSE_STYLED", "1") != "0"  # bf16 decoder: StyledConv output stages fused into the convolution / the Blur (off: the separate passes)


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1], demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample, blur_kernel=blur_kernel,
                                    demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def nhwc(self, x, style, noise=None):
        if x.dtype == torch.bfloat16 and FUSE_STYLED:
            out = self.conv.nhwc_styled(x, style, noise, self.noise.weight, self.activate.bias, self.activate.negative_slope, self.activate.scale)
            if out is not None:
                return out
        out = self.conv.nhwc(x, style)
        n, h, w, c = out.shape
        if noise is None:
            noise = torch.randn((n, h, w), device=out.device)
        else:
            noise = noise.expand(n, 1, h, w).reshape(n, h, w).contiguous()
        # NoiseInjection + FusedLeakyReLU in one pass (model.py:343-344)
        return FF.noise_bias_act(out, self.activate.bias, noise, self.noise.weight, self.activate.negative_slope, self.activate.scale)

    def forward(self, input, style, noise=None):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(input), style, noise))


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def nhwc(self, x, style, skip=None):
        res = self.upsample.nhwc(skip) if skip is not None else None
        if x.dtype == torch.bfloat16:  # bf16 decoder: dedicated bandwidth kernel, RGB / skip / bias stay fp32
            m = self.conv
            s = FF._Scale.apply(m.modulation(style), float(m.scale))
            return FF.torgb(x, m.weight.view(3, m.in_channel), s, self.bias.view(3), res)
        return self.conv.nhwc(x, style, bias=self.bias.view(3), residual=res)  # conv + bias + upsampled skip in the epilogue

    def forward(self, input, style, skip=None):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(input), style, None if skip is None else FF.to_nhwc(skip)))


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01, compute_dtype=torch.float32):
        super().__init__()
        # torch.bfloat16: the synthesis network keeps bf16 NHWC activations (fp32 accumulation, fp32 parameters / styles / noise /
        # RGB skip path) -- the "bf16 decoder" of BASELINE.json configs C3 / C5; the reference itself has no such switch
        self.compute_dtype = compute_dtype
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                         256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 2 ** 2, 2 ** 2, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, return_latents=False, return_features=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

        def lat(i):
            return latent[:, i].contiguous()

        out = FF.to_nhwc(self.input(latent))
        if self.compute_dtype == torch.bfloat16:
            out = out.to(torch.bfloat16)
        out = self.conv1.nhwc(out, lat(0), noise=noise[0])
        skip = self.to_rgb1.nhwc(out, lat(1))
        i = 1
        for conv1, conv2, noise1, noise2, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2], self.to_rgbs):
            out = conv1.nhwc(out, lat(i), noise=noise1)
            out = conv2.nhwc(out, lat(i + 1), noise=noise2)
            skip = to_rgb.nhwc(out, lat(i + 2), skip)
            i += 2
        image = FF.to_nchw(skip)
        if return_latents:
            return image, latent
        if return_features:
            return image, FF.to_nchw(out)
        return image, None
