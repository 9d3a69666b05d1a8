"""Host-side mirror of modules/psp/criteria/__init__.py (pSpLoss.__call__, criteria/__init__.py:44-99): ArcFace identity loss,
masked / plain L2, LPIPS-alex (masked), the reference-side LPIPS / L2, the W-norm term, and the VGG style / contextual terms (which the
reference computes for logging only -- they are never added to ``loss``, criteria/__init__.py:74-76,88-90).  The LPIPS / ArcFace
WEIGHTS are download-only: they are read from local files when present and stay randomly
initialised otherwise (weight parity unpinned)."""
from __future__ import annotations

import torch
from torch import nn

from .... import functional as FF
from ...loss import VGGLoss
from . import id_loss
from .lpips.lpips import LPIPS


class WNormLoss(nn.Module):
    def __init__(self, start_from_latent_avg=True):
        super().__init__()
        self.start_from_latent_avg = start_from_latent_avg

    def forward(self, latent, latent_avg=None):
        n = latent.shape[0]
        if self.start_from_latent_avg:
            latent = FF.add(latent, (-latent_avg).expand_as(latent).contiguous())
        ss = FF.sqsum_last(latent.reshape(n, -1))              # [N] squared Frobenius norms over (styles, 512)
        norms = FF.mul(ss, FF.rsqrt_eps(ss, 0.0))              # sqrt
        return FF.l1_loss(norms, torch.zeros_like(norms))      # norms >= 0: mean |.| = sum / N


class pSpLoss(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.id_lambda, self.lpips_lambda, self.l2_lambda, self.style_lambda = args.id_lambda, args.lpips_lambda, args.l2_lambda, args.style_lambda
        self.lpips_lambda_ref, self.l2_lambda_ref, self.cx_lambda = args.lpips_lambda_ref, args.l2_lambda_ref, args.cx_lambda
        self.w_norm_lambda = args.w_norm_lambda
        # True: loss_dict holds detached device tensors instead of python floats -- no host synchronisation inside __call__, so a
        # whole training step can be captured in a HIP graph; the caller converts after the replay (the reference converts in place)
        self.defer_logs = False
        if self.lpips_lambda > 0:  # the reference builds it on lpips_lambda alone and uses it for lpips_lambda_ref too (:29-30,82-85)
            self.lpips_loss = LPIPS(net_type="alex").eval()
        if self.id_lambda > 0:
            self.id_loss = id_loss.IDLoss().eval()
        if self.w_norm_lambda > 0:
            self.w_norm_loss = WNormLoss(start_from_latent_avg=args.start_from_latent_avg)
        if self.style_lambda > 0:
            self.vgg_loss = VGGLoss()

    def _log(self, v):
        return v.detach() if self.defer_logs else float(v.detach())

    def __call__(self, x, y, y_hat, latent, latent_avg=None, ref=None, mask=None):
        loss_dict, loss, id_logs = {}, 0.0, None
        m = mask.contiguous() if mask is not None else None  # [N,H,W]
        yh = FF.to_nhwc(y_hat)
        if self.id_lambda > 0:
            self.id_loss.defer_logs = self.defer_logs
            loss_id, sim_improvement, id_logs = self.id_loss(y_hat, y, x)
            loss_dict["loss_id"] = self._log(loss_id)
            loss_dict["id_improve"] = sim_improvement if self.defer_logs else float(sim_improvement)
            loss = loss_id * self.id_lambda
        if self.l2_lambda > 0:
            if m is not None:
                loss_l2 = FF.mse_loss(FF.mask_mul(yh, m, True), FF.mask_mul(FF.to_nhwc(y), m, True))
            else:
                loss_l2 = FF.mse_loss(yh, FF.to_nhwc(y))
            loss_dict["loss_l2"] = self._log(loss_l2)
            loss = loss + loss_l2 * self.l2_lambda
        if self.lpips_lambda > 0:
            if m is not None:
                loss_lpips = self.lpips_loss(FF.to_nchw(FF.mask_mul(yh, m, True)), FF.to_nchw(FF.mask_mul(FF.to_nhwc(y), m, True)))
            else:
                loss_lpips = self.lpips_loss(y_hat, y)
            loss_dict["loss_lpips"] = self._log(loss_lpips)
            loss = loss + loss_lpips * self.lpips_lambda
        if self.style_lambda > 0 and m is not None:
            with torch.no_grad():  # logged only in the reference (criteria/__init__.py:74-76)
                loss_dict["loss_style"] = self._log(self.vgg_loss(FF.to_nchw(FF.mask_mul(yh, m, True)), x, lossType="style") * self.style_lambda)
        if ref is not None:
            rf = FF.mask_mul(FF.to_nhwc(ref), m, False)
            yhm = FF.mask_mul(yh, m, False)
            if self.lpips_lambda_ref > 0:
                loss_lpips_ref = self.lpips_loss(FF.to_nchw(yhm), FF.to_nchw(rf))
                loss_dict["loss_lpips_ref"] = self._log(loss_lpips_ref)
                loss = loss + loss_lpips_ref * self.lpips_lambda_ref
            if self.l2_lambda_ref > 0:
                loss_l2_ref = FF.mse_loss(yhm, rf)
                loss_dict["loss_l2_ref"] = self._log(loss_l2_ref)
                loss = loss + loss_l2_ref * self.l2_lambda_ref
            if self.cx_lambda > 0:
                with torch.no_grad():  # logged only (criteria/__init__.py:88-90)
                    loss_dict["loss_context"] = self._log(self.vgg_loss(FF.to_nchw(yhm), FF.to_nchw(rf), lossType="contextual") * self.cx_lambda)
        if self.w_norm_lambda > 0 and latent_avg is not None:
            loss_w_norm = self.w_norm_loss(latent, latent_avg.to(latent.device))
            loss_dict["loss_w_norm"] = self._log(loss_w_norm)
            loss = loss + loss_w_norm * self.w_norm_lambda
        loss_dict["loss"] = self._log(loss) if torch.is_tensor(loss) else float(loss)
        return loss, loss_dict, id_logs
