"""Host-side mirror of modules/psp/psp.py (pSp, get_keys): GradualStyleEncoder -> W+ codes (+ latent_avg) -> StyleGAN2
Generator -> AdaptiveAvgPool2d(256), with the reference's forward signature.  ``load_weights`` follows the reference; when
no checkpoint is given and the default weight files are absent (no network here) the modules keep their random init."""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from ... import functional as FF
from ..pluralistic_model.base_function import _freeze
from . import model_paths
from .encoders import psp_encoders
from .stylegan2.model import Generator


def get_keys(d, name):
    if "state_dict" in d:
        d = d["state_dict"]
    return {k[len(name) + 1:]: v for k, v in d.items() if k[:len(name)] == name}


class pSp(nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.set_opts(opts)
        self.opts.n_styles = int(math.log(self.opts.output_size, 2)) * 2 - 2
        self.encoder = self.set_encoder()
        # opts.decoder_dtype ("bf16" / "fp32", this build's only extra option): bf16 synthesis network of configs C3 / C5
        dd = getattr(self.opts, "decoder_dtype", "fp32")
        dd = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}.get(dd, dd)
        self.decoder = Generator(self.opts.output_size, 512, 8, compute_dtype=dd)
        if not opts.train_decoder:
            _freeze(self.decoder)
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.load_weights()

    def set_encoder(self):
        if self.opts.encoder_type == "GradualStyleEncoder":
            return psp_encoders.GradualStyleEncoder(50, "ir_se", self.opts)
        raise Exception("{} is not a valid encoders".format(self.opts.encoder_type))

    def load_weights(self):
        self.latent_avg = None
        if getattr(self.opts, "pt_ckpt_path", None) is not None:
            ckpt = torch.load(self.opts.pt_ckpt_path, map_location="cpu", weights_only=True)
            self.encoder.load_state_dict(get_keys(ckpt, "encoder"), strict=False)
            self.decoder.load_state_dict(get_keys(ckpt, "decoder"), strict=True)
            self.__load_latent_avg(ckpt)
            return
        if not (os.path.exists(model_paths["ir_se50"]) and os.path.exists(getattr(self.opts, "stylegan_weights", None) or model_paths["stylegan_ffhq"])):
            print("pSp: pretrained weight files not found -> keeping the random initialisation")
            return
        self.encoder.load_state_dict(torch.load(model_paths["ir_se50"], map_location="cpu", weights_only=True), strict=False)
        ckpt = torch.load(self.opts.stylegan_weights if getattr(self.opts, "stylegan_weights", None) else model_paths["stylegan_ffhq"], map_location="cpu", weights_only=True)
        self.decoder.load_state_dict(ckpt["g_ema"], strict=False)
        self.__load_latent_avg(ckpt, repeat=1 if self.opts.learn_in_w else self.opts.n_styles)

    def forward(self, x, ref=None, src_mask=None, resize=True, latent_mask=None, input_code=False, randomize_noise=True,
                inject_latent=None, return_latents=False, alpha=None):
        if input_code:
            codes = x
        else:
            codes = self.encoder(x, ref=ref, mask=src_mask)
            if self.opts.start_from_latent_avg and self.latent_avg is not None:
                avg = self.latent_avg.to(codes.device)
                codes = FF.add(codes, (avg.repeat(codes.shape[0], 1) if self.opts.learn_in_w else avg.repeat(codes.shape[0], 1, 1)).contiguous())
        if latent_mask is not None:
            for i in latent_mask:
                if inject_latent is not None:
                    codes[:, i] = alpha * inject_latent[:, i] + (1 - alpha) * codes[:, i] if alpha is not None else inject_latent[:, i]
                else:
                    codes[:, i] = 0
        images, result_latent = self.decoder([codes], input_is_latent=not input_code, randomize_noise=randomize_noise,
                                             return_latents=return_latents)
        if resize:  # self.face_pool = AdaptiveAvgPool2d((256, 256)): any decoder output size (replication below 256, psp.py:33,113-114)
            images = FF.to_nchw(FF.adaptive_avg_pool(FF.to_nhwc(images), 256, 256))
        return (images, result_latent) if return_latents else images

    def set_opts(self, opts):
        self.opts = opts

    def __load_latent_avg(self, ckpt, repeat=None):
        if "latent_avg" in ckpt:
            self.latent_avg = ckpt["latent_avg"]
            if repeat is not None:
                self.latent_avg = self.latent_avg.repeat(repeat, 1)
        else:
            self.latent_avg = None
