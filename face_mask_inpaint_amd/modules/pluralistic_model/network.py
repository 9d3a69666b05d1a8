"""Host-side mirror of modules/pluralistic_model/network.py: define_e / define_g / define_d and the
ResEncoder / ResGenerator / ResDiscriminator networks with the reference's constructor arguments, attribute
names (``block0``, ``encoder{i}``, ``infer_prior{i}``, ``prior`` / ``posterior``, ``generator``, ``decoder{i}``,
``out{i}``, ``attn{i}``, ``block1``, ``conv``) and forward signatures; all arithmetic runs in the HIP kernels.
PatchDiscriminator (``--disc_model_type PatchDis``, network.py:373-430) is mirrored too.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import functional as FF
from ...weights import weight_scope
from .base_function import (Auto_Attn, Output, ResBlock, ResBlockDecoder, ResBlockEncoderOptimized, _conv, get_nonlinearity_layer,
                            get_norm_layer, init_net, _slope)
from .external_function import SpectralNorm, run_conv


def define_e(encoder_type="src", input_nc=3, ngf=64, z_nc=512, img_f=512, L=6, layers=5, norm="none", activation="ReLU",
             use_spect=True, use_coord=False, init_type="orthogonal", gpu_ids=[]):
    net = ResEncoder(input_nc, ngf, z_nc, img_f, L, layers, norm, activation, use_spect, use_coord, encoder_type)
    return init_net(net, init_type, activation, gpu_ids)


def define_g(output_nc=3, ngf=64, z_nc=512, img_f=512, L=1, layers=5, norm="instance", activation="ReLU", use_spect=True,
             use_coord=False, use_attn=True, init_type="orthogonal", gpu_ids=[]):
    net = ResGenerator(output_nc, ngf, z_nc, img_f, L, layers, norm, activation, use_spect, use_coord, use_attn)
    return init_net(net, init_type, activation, gpu_ids)


def define_d(input_nc=3, ndf=64, img_f=512, layers=6, norm="none", activation="LeakyReLU", use_spect=True, use_coord=False,
             use_attn=True, model_type="ResDis", init_type="orthogonal", gpu_ids=[]):
    if model_type == "ResDis":
        net = ResDiscriminator(input_nc, ndf, img_f, layers, norm, activation, use_spect, use_coord, use_attn)
    elif model_type == "PatchDis":
        net = PatchDiscriminator(input_nc, ndf, img_f, layers, norm, activation, use_spect, use_coord, use_attn)
    else:
        raise NotImplementedError("model_type %s" % model_type)
    return init_net(net, init_type, activation, gpu_ids)


class ResEncoder(nn.Module):
    """network.py:76-178.  forward(img) -> ([mu, softplus(std)], feature); the fused training path uses
    ``nhwc_raw`` which returns the un-split distribution head so that split/softplus/rsample happen in one kernel."""

    def __init__(self, input_nc=3, ngf=64, z_nc=128, img_f=1024, L=6, layers=6, norm="none", activation="ReLU",
                 use_spect=True, use_coord=False, encoder_type="src"):
        super().__init__()
        self.layers, self.z_nc, self.L = layers, z_nc, L
        self.ecnoder_type = encoder_type  # (sic) attribute name of the reference
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.block0 = ResBlockEncoderOptimized(input_nc, ngf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ngf)
            block = ResBlock(ngf * mult_prev, ngf * mult, ngf * mult_prev, norm_layer, nonlinearity,
                             "none" if i % 2 == 0 else "down", use_spect, use_coord)
            setattr(self, "encoder" + str(i), block)
        if encoder_type == "src":
            for i in range(self.L):
                setattr(self, "infer_prior" + str(i),
                        ResBlock(ngf * mult, ngf * mult, ngf * mult, norm_layer, nonlinearity, "none", use_spect, use_coord))
            self.prior = ResBlock(ngf * mult, 2 * z_nc, ngf * mult, norm_layer, nonlinearity, "none", use_spect, use_coord)
        elif encoder_type == "ref":
            self.posterior = ResBlock(ngf * mult, 2 * z_nc, ngf * mult, norm_layer, nonlinearity, "none", use_spect, use_coord)
        else:
            raise NotImplementedError(encoder_type)

    def nhwc_raw(self, img):
        """returns (o [N,h,w,2*z_nc] = un-split (mu, raw std), feature [N,h,w,C])"""
        with weight_scope(self):
            out = self.block0.nhwc(img)
            for i in range(self.layers - 1):
                out = getattr(self, "encoder" + str(i)).nhwc(out)
            enc = out
            if self.ecnoder_type == "src":
                for i in range(self.L):
                    enc = getattr(self, "infer_prior" + str(i)).nhwc(enc)
                o = self.prior.nhwc(enc)
            else:
                o = self.posterior.nhwc(enc)
            return o, out

    def forward(self, img):
        o, feat = self.nhwc_raw(FF.to_nhwc(img))
        mu = o[..., : self.z_nc].contiguous()
        std = FF._Softplus.apply(o[..., self.z_nc:].contiguous())
        return [FF.to_nchw(mu), FF.to_nchw(std)], FF.to_nchw(feat)


class ResGenerator(nn.Module):
    """network.py:181-307."""

    def __init__(self, output_nc=3, ngf=64, z_nc=128, img_f=1024, L=1, layers=6, norm="batch", activation="ReLU",
                 use_spect=True, use_coord=False, use_attn=False):
        super().__init__()
        self.layers, self.L, self.use_attn = layers, L, use_attn
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        mult = min(2 ** (layers - 1), img_f // ngf)
        ch = int(ngf * mult)
        self.generator = ResBlock(z_nc, ch, ch, None, nonlinearity, "none", use_spect, use_coord)
        for i in range(self.L):
            setattr(self, "generator" + str(i), ResBlock(ch, ch, ch, None, nonlinearity, "none", use_spect, use_coord))
        for i in range(layers):
            mult_prev = mult
            mult = min(2 ** (layers - i - 1), img_f // ngf)
            prev_ch, ch = int(ngf * mult_prev), int(ngf * mult)
            setattr(self, "decoder" + str(i), ResBlockDecoder(prev_ch, ch, ch, norm_layer, nonlinearity, use_spect, use_coord))
            if i > layers - 2:
                setattr(self, "out" + str(i), Output(ch, output_nc, 3, None, nonlinearity, use_spect, use_coord))
            if i == 1 and use_attn:
                setattr(self, "attn" + str(i), Auto_Attn(ch, None))

    def nhwc(self, encoded, z=None):
        with weight_scope(self):
            if z is not None:
                f = self.generator.nhwc(z)
                for i in range(self.L):
                    f = getattr(self, "generator" + str(i)).nhwc(f)
                out = FF.add(encoded, f)
            else:
                out = encoded
            output = None
            for i in range(self.layers):
                out = getattr(self, "decoder" + str(i)).nhwc(out)
                if i == 1 and self.use_attn:
                    out = getattr(self, "attn" + str(i)).nhwc(out)
                if i > self.layers - 2:
                    # the reference also builds cat([out, output]) here, which nothing reads (network.py:272)
                    output = getattr(self, "out" + str(i)).nhwc(out)
            return output

    def forward(self, encoded, z=None, f_e=None, mask=None):
        if f_e is not None or mask is not None:
            raise NotImplementedError("f_e / mask are never passed by the reference's callers")
        return FF.to_nchw(self.nhwc(FF.to_nhwc(encoded), None if z is None else FF.to_nhwc(z)))

    def get_z(self, src_distribution, ref_distribution, return_zq=False, mask=None, eps=None):
        """network.py:275-307 on NCHW-shaped (mu, sigma) pairs.  ``eps = (eps_p, eps_q)`` injects the two
        standard-normal draws (posterior first, as rsample is called in the reference); default: fresh draws."""
        p_mu, p_sigma = ref_distribution
        q_mu, q_sigma = src_distribution
        if eps is None:
            eps = (torch.randn_like(p_mu), torch.randn_like(q_mu))
        z_p = FF._MulAdd.apply(p_sigma.contiguous(), eps[0].contiguous(), p_mu.contiguous())
        z_q = FF._MulAdd.apply(q_sigma.contiguous(), eps[1].contiguous(), q_mu.contiguous())
        if return_zq:
            return z_q
        return torch.cat([z_q, z_p], dim=1)


class ResDiscriminator(nn.Module):
    """network.py:310-370."""

    def __init__(self, input_nc=3, ndf=64, img_f=1024, layers=6, norm="none", activation="LeakyReLU", use_spect=True,
                 use_coord=False, use_attn=True):
        super().__init__()
        self.layers, self.use_attn = layers, use_attn
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.nonlinearity = nonlinearity
        self.block0 = ResBlockEncoderOptimized(input_nc, ndf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ndf)
            if i == 2 and use_attn:
                setattr(self, "attn" + str(i), Auto_Attn(ndf * mult_prev, norm_layer))
            setattr(self, "encoder" + str(i),
                    ResBlock(ndf * mult_prev, ndf * mult, ndf * mult_prev, norm_layer, nonlinearity, "down", use_spect, use_coord))
        self.block1 = ResBlock(ndf * mult, ndf * mult, ndf * mult, norm_layer, nonlinearity, "none", use_spect, use_coord)
        self.conv = SpectralNorm(nn.Conv2d(ndf * mult, 1, 3))
        self._slope = _slope(nonlinearity)

    def nhwc(self, x):
        with weight_scope(self):
            out = self.block0.nhwc(x)
            for i in range(self.layers - 1):
                if i == 2 and self.use_attn:
                    out = getattr(self, "attn" + str(i)).nhwc(out)
                out = getattr(self, "encoder" + str(i)).nhwc(out)
            out = self.block1.nhwc(out)
            return run_conv(_conv(self.conv), FF.leaky_relu(out, self._slope))

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))


class PatchDiscriminator(nn.Module):
    """network.py:373-430 (``--disc_model_type PatchDis``): ``layers`` 4x4 stride-2 SpectralNorm convs + two 4x4 stride-1 ones,
    LeakyReLU in between, no bias, no norm layer (the reference builds ``norm_layer`` and never uses it).  ``model.N`` keys as in
    the reference's nn.Sequential."""

    def __init__(self, input_nc=3, ndf=64, img_f=512, layers=3, norm="batch", activation="LeakyReLU", use_spect=True, use_coord=False,
                 use_attn=False):
        super().__init__()
        from .base_function import coord_conv

        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self._slope = _slope(nonlinearity)
        kwargs = {"kernel_size": 4, "stride": 2, "padding": 1, "bias": False}
        sequence = [coord_conv(input_nc, ndf, use_spect, use_coord, **kwargs), nonlinearity]
        mult, i = 1, 0
        for i in range(1, layers):
            mult_prev = mult
            mult = min(2 ** i, img_f // ndf)
            sequence += [coord_conv(ndf * mult_prev, ndf * mult, use_spect, use_coord, **kwargs), nonlinearity]
        mult_prev = mult
        mult = min(2 ** i, img_f // ndf)
        kwargs = {"kernel_size": 4, "stride": 1, "padding": 1, "bias": False}
        sequence += [coord_conv(ndf * mult_prev, ndf * mult, use_spect, use_coord, **kwargs), nonlinearity,
                     coord_conv(ndf * mult, 1, use_spect, use_coord, **kwargs)]
        self.model = nn.Sequential(*sequence)

    def nhwc(self, x):
        with weight_scope(self):
            for m in self.model:
                x = FF.leaky_relu(x, self._slope) if isinstance(m, (nn.LeakyReLU, nn.ReLU)) else run_conv(_conv(m), x)
            return x

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))
