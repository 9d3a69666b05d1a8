"""Host-side mirror of modules/evaluations/ssim.py (gaussian / create_window / ssim / SSIM, same signatures): the
11-tap sigma-1.5 window, C1 = 0.01^2, C2 = 0.03^2, zero-padded filtering -- one HIP kernel (fmi_ssim_f32) computes the
five filtered maps, the ssim map and its mean.  Metric only (the reference never back-propagates through it).
This is the in-repo SSIM; pytorch_msssim (used by the trainers' eval loops) is a different, absent package."""
from __future__ import annotations

from math import exp

import torch

from ... import _lib
from ...functional import _chk, _p, _st


def gaussian(window_size, sigma):
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    return w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    a, b = img1.contiguous(), img2.contiguous()
    _chk(a, b)
    n, c, h, w = a.shape
    g = gaussian(window_size, 1.5).to(a.device)
    out = torch.zeros(1 if size_average else n, device=a.device, dtype=torch.float32)
    _lib.lib().ssim_f32(_p(a), _p(b), _p(g), window_size, n * c, h, w, n * c if size_average else c, _p(out), _st())
    out = out / float((n * c if size_average else c) * h * w)
    return out[0] if size_average else out


class SSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1
        self.window = create_window(window_size, self.channel)

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
