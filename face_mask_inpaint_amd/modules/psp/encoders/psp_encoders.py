"""Host-side mirror of modules/psp/encoders/psp_encoders.py:13-152 (GradualStyleBlock, GradualStyleEncoder): IR-SE50
feature pyramid on source and reference, example-guided attention at 16^2 / 32^2, mask blend at 64^2, 14-18 map2style heads.
BackboneEncoderUsingLastLayerIntoW(Plus) are not the default encoder and are absent."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn
from torch.nn import BatchNorm2d, Conv2d, Module, PReLU, Sequential

from .... import functional as FF
from ....weights import weight_scope
from ...example_guided_att import ExampleGuidedAttention
from ...pluralistic_model.external_function import run_conv
from ..stylegan2.model import EqualLinear
from . import helpers
from .helpers import batch_norm, bottleneck_IR, bottleneck_IR_SE, get_blocks


class GradualStyleBlock(Module):
    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c = out_c
        self.spatial = spatial
        num_pools = int(np.log2(spatial))
        modules = [Conv2d(in_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        for _ in range(num_pools - 1):
            modules += [Conv2d(out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*modules)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def nhwc(self, x):
        with weight_scope(self):
            for m in self.convs:
                x = run_conv(m, x) if isinstance(m, Conv2d) else FF.leaky_relu(x, m.negative_slope)
            return self.linear(x.reshape(-1, self.out_c))

    def forward(self, x):
        return self.nhwc(FF.to_nhwc(x))


class GradualStyleEncoder(Module):
    def __init__(self, num_layers, mode="ir", opts=None, *, _widths=(64, 64, 128, 256, 512), _spatial=(16, 32, 64)):
        """``_widths`` (stem + the four stage depths) and ``_spatial`` (map sizes feeding the coarse / middle / fine heads)
        default to the reference's hard-coded IR-50 / 256x256 geometry; tests shrink them to keep fixtures small."""
        super().__init__()
        assert num_layers in [50, 100, 152], "num_layers should be 50,100, or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        unit_module = bottleneck_IR if mode == "ir" else bottleneck_IR_SE
        w0, w1, w2, w3, w4 = _widths
        self.input_layer = Sequential(Conv2d(3, w0, (3, 3), 1, 1, bias=False), BatchNorm2d(w0), PReLU(w0))
        scale = {64: w1, 128: w2, 256: w3, 512: w4}
        self.body = Sequential(*[unit_module(w0 if i == 0 and j == 0 else scale[b.in_channel], scale[b.depth], b.stride)
                                 for i, block in enumerate(get_blocks(num_layers)) for j, b in enumerate(block)])
        self.styles = nn.ModuleList()
        self.style_count = opts.n_styles
        self.coarse_ind = 3
        self.middle_ind = 7
        for i in range(self.style_count):
            self.styles.append(GradualStyleBlock(w4, w4, _spatial[0] if i < self.coarse_ind else (_spatial[1] if i < self.middle_ind else _spatial[2])))
        self.latlayer1 = nn.Conv2d(w3, w4, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(w2, w4, kernel_size=1, stride=1, padding=0)
        # opts.encoder_dtype ("fp32" default / "bf16", this build's option like opts.decoder_dtype): the 24 IR-SE blocks keep bf16 NHWC
        # activations between the bf16 MFMA convolutions (fp32 accumulation, fp32 parameters / BatchNorm statistics / SE gates and all
        # parameter gradients); the C = 3 stem, the attention blocks, the FPN and the style heads stay fp32
        dt = getattr(opts, "encoder_dtype", "fp32")
        self.body_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}.get(dt, dt)
        self.use_attention = opts.use_attention
        if opts.use_attention:
            self.attention1 = ExampleGuidedAttention(w4, out_channels=w4)
            self.attention2 = ExampleGuidedAttention(w3, out_channels=w3)

    def _pyramid(self, x):
        x = run_conv(self.input_layer[0], x)
        x = FF.prelu(batch_norm(self.input_layer[1], x), self.input_layer[2].weight)
        b16 = self.body_dtype == torch.bfloat16 and self.training
        if getattr(self, "_fmi_body_b16", None) != b16:  # the body's convolutions need no fp32 piece images while they run on bf16 activations
            for m in self.body.modules():
                if isinstance(m, nn.Conv2d):
                    object.__setattr__(m, "_fmi_no_w3", b16)
            object.__setattr__(self, "_fmi_body_b16", b16)
        if b16:
            x = x.to(torch.bfloat16)
        taps = {}
        for i, l in enumerate(self.body):
            x = l.nhwc(x)
            if i in (6, 20, 23):
                taps[i] = x.float() if b16 else x
        return taps[6], taps[20], taps[23]  # [N,64,64,128], [N,32,32,256], [N,16,16,512]

    @staticmethod
    def _upsample_add(x, y):
        return FF.add(FF.resize_bilinear(x, y.shape[1], y.shape[2]), y)

    @staticmethod
    def _blend(m, r, c):
        """m*r + (1-m)*c with m [N,H,W] (psp_encoders.py:135-138)"""
        return FF.add(FF.mask_mul(r, m, False), FF.mask_mul(c, m, True))

    def forward(self, x, ref=None, mask=None):
        with weight_scope(self):
            if ref is not None:
                assert mask is not None, "ref and mask should both be provided"
                # psp_encoders.py:101-125 runs input_layer + body on x and then AGAIN on ref (shared weights).  Here both go through
                # as ONE batch of 2N -- every convolution launch has twice the rows (the 256-channel 32^2 stage is 256 tiles = one
                # workgroup per CU at N = 16) and half the launches -- while training-mode BatchNorm keeps per-part statistics and
                # updates its running buffers part by part (helpers.BN_GROUPS), so every value equals the two-pass form
                nb = x.shape[0]
                helpers.BN_GROUPS[0] = 2
                try:
                    t1, t2, t3 = self._pyramid(torch.cat([FF.to_nhwc(x), FF.to_nhwc(ref)], dim=0))
                finally:
                    helpers.BN_GROUPS[0] = 1
                (c1, r1), (c2, r2), (c3, r3) = FF.split_batch(t1, nb), FF.split_batch(t2, nb), FF.split_batch(t3, nb)
                m = mask.contiguous().unsqueeze(-1)  # [N,H,W,1]
                n = m.shape[0]
                mask_3 = FF.resize_bilinear(m, r3.shape[1], r3.shape[2]).view(n, r3.shape[1], r3.shape[2])
                mask_2 = FF.resize_bilinear(m, r2.shape[1], r2.shape[2]).view(n, r2.shape[1], r2.shape[2])
                mask_1 = FF.resize_bilinear(m, r1.shape[1], r1.shape[2]).view(n, r1.shape[1], r1.shape[2])
                if self.use_attention:
                    c3 = self.attention1.nhwc(mask_3, c3, r3)
                    c2 = self.attention2.nhwc(mask_2, c2, r2)
                else:
                    c3 = self._blend(mask_3, r3, c3)
                    c2 = self._blend(mask_2, r2, c2)
                c1 = self._blend(mask_1, r1, c1)
            else:
                c1, c2, c3 = self._pyramid(FF.to_nhwc(x))
            latents = [self.styles[j].nhwc(c3) for j in range(self.coarse_ind)]
            p2 = self._upsample_add(c3, run_conv(self.latlayer1, c2))
            latents += [self.styles[j].nhwc(p2) for j in range(self.coarse_ind, self.middle_ind)]
            p1 = self._upsample_add(p2, run_conv(self.latlayer2, c1))
            latents += [self.styles[j].nhwc(p1) for j in range(self.middle_ind, self.style_count)]
            return torch.stack(latents, dim=1)
