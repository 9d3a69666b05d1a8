"""Host-side mirror of modules/unet/unet_model.py: UNet(n_channels, n_classes, bilinear) -- the mask detector's network."""
from __future__ import annotations

from torch import nn

from ... import functional as FF
from ...weights import weight_scope
from .unet_parts import DoubleConv, Down, OutConv, Up


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=True):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        factor = 2 if bilinear else 1
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)

    def nhwc(self, x):
        with weight_scope(self):
            x1 = self.inc.nhwc(x)
            x2 = self.down1.nhwc(x1)
            x3 = self.down2.nhwc(x2)
            x4 = self.down3.nhwc(x3)
            x5 = self.down4.nhwc(x4)
            x = self.up1.nhwc(x5, x4)
            x = self.up2.nhwc(x, x3)
            x = self.up3.nhwc(x, x2)
            x = self.up4.nhwc(x, x1)
            return self.outc.nhwc(x)

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))
