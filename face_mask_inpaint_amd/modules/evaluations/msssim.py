"""SSIM / MS-SSIM with the call signatures the reference's trainers use from ``pytorch_msssim``
(``SSIM(data_range=1, size_average=True, channel=3)``, ``MS_SSIM(...)``: train_reference_fill.py:17,207-209,252-257, train_psp.py:16,176-178,
PICNet_inference.py:8,130-131, dataloader.py:16,165).  That package is a third-party dependency, absent from /root/reference and from this
image, so what is implemented here is its published definition (Wang, Simoncelli, Bovik 2003: five scales, weights 0.0448 / 0.2856 /
0.3001 / 0.2363 / 0.1333; 11-tap Gaussian of sigma 1.5 applied WITHOUT padding, K1 = 0.01, K2 = 0.03, 2 x 2 mean pooling between scales
with odd sizes zero-padded, contrast-structure terms clamped at zero) -- **parity unpinned**: no reference fixture can exist.  It differs
from the in-repo ``modules/evaluations/ssim.py`` (zero-padded filtering), which is mirrored next door in ``ssim.py``.

The per-plane means of the ssim map and of its cs factor come from one kernel per scale (fmi_ssim_valid_f32, reproducible fixed-order
reduction); the final product over at most 5 x N x C numbers is formed on the host in float64.  Metric only: no gradient."""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib
from ...functional import FmiError, _chk, _p, _st
from .ssim import gaussian

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _plane_stats(a, b, win, c1, c2):
    """[N*C][2] = (mean ssim, mean cs) per plane of NCHW tensors"""
    n, c, h, w = a.shape
    out = torch.empty((n * c, 2), device=a.device, dtype=torch.float32)
    part = torch.empty(n * c * 64 * 2, device=a.device, dtype=torch.float64)
    _lib.lib().ssim_valid_f32(_p(a), _p(b), _p(win), win.numel(), n * c, h, w, c1, c2, _p(out), C.c_void_p(part.data_ptr()), part.numel(), _st())
    return out


def _check(x, y, win_size):
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("expected two NCHW batches of one shape")
    if win_size % 2 != 1:
        raise ValueError("Window size should be odd.")
    a, b = x.contiguous(), y.contiguous()
    _chk(a, b)
    return a, b


def ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, K=(0.01, 0.03), nonnegative_ssim=False):
    a, b = _check(X, Y, win_size)
    if min(a.shape[2:]) < win_size:
        raise FmiError("image smaller than the SSIM window")
    win = gaussian(win_size, win_sigma).to(a.device)
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    per = _plane_stats(a, b, win, c1, c2)[:, 0].double().view(a.shape[0], a.shape[1])
    if nonnegative_ssim:
        per = per.clamp_min(0)
    return (per.mean() if size_average else per.mean(1)).float()


def ms_ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, weights=None, K=(0.01, 0.03)):
    a, b = _check(X, Y, win_size)
    weights = MS_WEIGHTS if weights is None else tuple(float(v) for v in weights)
    levels = len(weights)
    if min(a.shape[2:]) <= (win_size - 1) * 2 ** (levels - 1):
        raise AssertionError("Image size should be larger than %d due to the %d downsamplings in ms-ssim" % ((win_size - 1) * 2 ** (levels - 1), levels - 1))
    win = gaussian(win_size, win_sigma).to(a.device)
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    n, c = a.shape[:2]
    lib = _lib.lib()
    stats = []
    for lv in range(levels):
        stats.append(_plane_stats(a, b, win, c1, c2))
        if lv < levels - 1:
            h, w = a.shape[2:]
            ph, pw = h % 2, w % 2
            oh, ow = (h + 2 * ph - 2) // 2 + 1, (w + 2 * pw - 2) // 2 + 1
            na, nb = (torch.empty((n, c, oh, ow), device=a.device, dtype=torch.float32) for _ in range(2))
            lib.avgpool2_pad_f32(_p(a), _p(na), n * c, h, w, ph, pw, _st())
            lib.avgpool2_pad_f32(_p(b), _p(nb), n * c, h, w, ph, pw, _st())
            a, b = na, nb
    st = torch.stack(stats).double().cpu()  # [levels][N*C][2]: one small device -> host read
    terms = torch.cat([st[:-1, :, 1], st[-1:, :, 0]]).clamp_min(0)  # cs of the first levels, ssim of the coarsest
    val = torch.prod(terms ** torch.tensor(weights, dtype=torch.float64).view(-1, 1), dim=0).view(n, c)
    res = val.mean() if size_average else val.mean(1)
    return res.float().to(X.device)


class SSIM(torch.nn.Module):
    def __init__(self, data_range=255, size_average=True, win_size=11, win_sigma=1.5, channel=3, spatial_dims=2, K=(0.01, 0.03), nonnegative_ssim=False):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("2-d images only")
        self.cfg = dict(data_range=data_range, size_average=size_average, win_size=win_size, win_sigma=win_sigma, K=K, nonnegative_ssim=nonnegative_ssim)

    def forward(self, X, Y):
        return ssim(X, Y, **self.cfg)


class MS_SSIM(torch.nn.Module):
    def __init__(self, data_range=255, size_average=True, win_size=11, win_sigma=1.5, channel=3, spatial_dims=2, weights=None, K=(0.01, 0.03)):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("2-d images only")
        self.cfg = dict(data_range=data_range, size_average=size_average, win_size=win_size, win_sigma=win_sigma, weights=weights, K=K)

    def forward(self, X, Y):
        return ms_ssim(X, Y, **self.cfg)
