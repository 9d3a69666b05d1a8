"""Host-side mirror of modules/model.py: ReferenceFill (model.py:15-112) and scale_img (model.py:10-12)."""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as FF
from ..weights import weight_scope
from .example_guided_att import ExampleGuidedAttention
from .pluralistic_model import network


def scale_img(img, size):
    """bilinear, align_corners=True on an NCHW-shaped tensor (model.py:10-12)."""
    return FF.to_nchw(FF.resize_bilinear(FF.to_nhwc(img), int(size[0]), int(size[1])))


class ReferenceFill(nn.Module):
    def __init__(self, mask_detector, encoder_params, decoder_params, use_att=True, out_size=(256, 256)):
        super().__init__()
        self.mask_detector = mask_detector
        self.encoder_type = encoder_params.pop("type")
        if self.encoder_type == "drn":  # model.py:47-59
            from .drn import drn_c_42

            self.src_encoder = drn_c_42(pretrained=False, out_map=True)
            self.src_encoder.fc = torch.nn.Conv2d(self.src_encoder.out_dim, encoder_params["img_f"], kernel_size=1, stride=1, padding=0, bias=True)
            self.ref_encoder = drn_c_42(pretrained=False, out_map=True)
            self.ref_encoder.fc = torch.nn.Conv2d(self.ref_encoder.out_dim, encoder_params["img_f"], kernel_size=1, stride=1, padding=0, bias=True)
        elif self.encoder_type == "pluralistic":
            self.src_encoder = network.define_e(**encoder_params, encoder_type="src")
            self.ref_encoder = network.define_e(**encoder_params, encoder_type="ref")
        else:
            raise NotImplementedError
        self.decoder = network.define_g(**decoder_params)
        self.use_att = use_att
        if use_att:
            self.attention = ExampleGuidedAttention(encoder_params["img_f"])
        self.pool = nn.AdaptiveAvgPool2d(out_size)
        self._out_size = tuple(out_size)

    def forward(self, src_image, ref_image, src_mask=None, resize=True, no_prior=False, eps=None):
        """src_image / ref_image [N,3,H,W]; src_mask [N,H,W] float {0,1}.  ``eps = (eps_p, eps_q)`` optionally injects
        the two standard-normal draws of get_z ([N, z_nc, h, w], posterior first); default: fresh torch.randn draws.
        Variants of model.py:97-112: ``use_att=False`` blends the two feature maps with the rescaled mask and decodes from z_q
        alone; ``no_prior=True`` (what PICNet_inference.py:107 passes for --old_model) decodes without z and rescales the image to
        218 x 178 with scale_img instead of pooling."""
        if src_mask is None:
            if self.mask_detector is None:
                raise ValueError("src_mask is required when no mask_detector is attached")
            # model.py:86 feeds the detector's boolean [N, 2, H, W] output to scale_img, which fails in the reference as well
            # (F.interpolate of a 5-D bool tensor); the working call sites pass argmax masks (PICNet_inference.py:100-101)
            src_mask = self.mask_detector.predict_mask(src_image)
        skip = ()
        skip_z = no_prior or self.encoder_type == "drn"  # model.py:103-104: a DRN encoder has no latent distribution
        if skip_z:  # the decoder's latent blocks do not run without z (network.py:253-260)
            skip = tuple(getattr(self.decoder, n) for n in ["generator"] + ["generator%d" % i for i in range(self.decoder.L)])
        with weight_scope(self, skip):
            src = FF.to_nhwc(src_image)
            ref = FF.to_nhwc(ref_image)
            if self.encoder_type == "drn":
                (src_feat, _), (ref_feat, _) = self.src_encoder.nhwc(src), self.ref_encoder.nhwc(ref)
                o_src = o_ref = None
            else:
                o_src, src_feat = self.src_encoder.nhwc_raw(src)
                o_ref, ref_feat = self.ref_encoder.nhwc_raw(ref)
            n, fh, fw, _ = src_feat.shape
            m = FF.resize_bilinear(src_mask.contiguous().unsqueeze(-1), fh, fw).view(n, fh, fw)
            if self.use_att:
                enc = self.attention.nhwc(m, src_feat, ref_feat)
            else:  # (1 - m) * src + m * ref  (model.py:100-101)
                enc = FF.add(FF.mask_mul(src_feat, m, True), FF.mask_mul(ref_feat, m, False))
            if skip_z:
                img = self.decoder.nhwc(enc, None)
            else:
                z_nc = o_src.shape[-1] // 2
                if eps is None:
                    eps_p = torch.randn((n, fh, fw, z_nc), device=src.device)
                    eps_q = torch.randn((n, fh, fw, z_nc), device=src.device)
                else:
                    eps_p, eps_q = FF.to_nhwc(eps[0]), FF.to_nhwc(eps[1])
                z = FF.vae_sample(o_src, o_ref, eps_q, eps_p)  # [z_q, z_p] along channels
                if not self.use_att:  # get_z(..., return_zq=True): the prior sample alone (model.py:106)
                    z = FF.slice_channels(z, 0, z_nc)
                img = self.decoder.nhwc(enc, z)
            if resize:
                if no_prior:
                    img = FF.resize_bilinear(img, 218, 178)  # scale_img(dec_image, (218, 178)), model.py:109-110
                else:
                    img = FF.adaptive_avg_pool(img, *self._out_size)
            return FF.to_nchw(img)
