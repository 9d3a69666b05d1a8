"""Host-side mirror of modules/unet/unet_parts.py (DoubleConv, Down, Up, OutConv) with the reference's parameter names
(``double_conv.N``, ``maxpool_conv.1``, ``up`` / ``conv``); forward on the HIP kernels in NHWC."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import functional as FF
from ...weights import weight_scope
from ..pluralistic_model.external_function import run_conv
from ..psp.encoders.helpers import batch_norm


class _Nhwc(nn.Module):
    def forward(self, *xs):
        return FF.to_nchw(self.nhwc(*[FF.to_nhwc(x) for x in xs]))


class DoubleConv(_Nhwc):
    """(convolution => [BN] => ReLU) * 2   (unet_parts.py:8-27)"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1), nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def nhwc(self, x):
        with weight_scope(self):
            for conv, bn in ((self.double_conv[0], self.double_conv[1]), (self.double_conv[3], self.double_conv[4])):
                x = FF.leaky_relu(batch_norm(bn, run_conv(conv, x)), 0.0)
            return x


class Down(_Nhwc):
    """MaxPool2d(2) then DoubleConv (unet_parts.py:30-42)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def nhwc(self, x):
        with weight_scope(self):
            return self.maxpool_conv[1].nhwc(FF.max_pool(x, 2, 2))


class Up(_Nhwc):
    """bilinear x2 (align_corners=True) or ConvTranspose2d(k2, s2), pad to the skip's size, concat [skip, up], DoubleConv
    (unet_parts.py:45-72)"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def nhwc(self, x1, x2):
        with weight_scope(self):
            if isinstance(self.up, nn.Upsample):
                x1 = FF.resize_bilinear(x1, 2 * x1.shape[1], 2 * x1.shape[2])
            else:
                raise NotImplementedError("Up(bilinear=False): every caller of the reference builds MaskDetector(bilinear=True)")
            dy, dx = x2.shape[1] - x1.shape[1], x2.shape[2] - x1.shape[2]
            if dy or dx:  # odd input sizes only: zero border (torch plumbing, copies a few rows)
                x1 = torch.nn.functional.pad(x1, [0, 0, dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
            return self.conv.nhwc(FF.cat_channels(x2, x1))


class OutConv(_Nhwc):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def nhwc(self, x):
        with weight_scope(self):
            return run_conv(self.conv, x)
