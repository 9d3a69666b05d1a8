"""Host-side mirror of modules/psp/criteria/lpips/networks.py: the AlexNet trunk of LPIPS (torchvision ``alexnet().features``
restated layer by layer -- torchvision is absent offline --, ``target_layers`` [2, 5, 8, 10, 12], ``n_channels_list``
[64, 192, 384, 256, 256]) and the 1x1 linear heads, with the reference's parameter names (``layers.N.weight``, ``mean`` / ``std``
buffers, ``lin.N.1.weight``).  'squeeze' / 'vgg' trunks are never selected by the reference's callers (criteria/__init__.py:30)."""
from __future__ import annotations

from itertools import chain
from typing import Sequence

import torch
import torch.nn as nn

from ..... import functional as FF
from .....weights import weight_scope
from ....pluralistic_model.external_function import run_conv
from .utils import normalize_activation


def get_network(net_type: str):
    if net_type == "alex":
        return AlexNet()
    raise NotImplementedError("LPIPS trunk %r: the reference's callers only build 'alex' (criteria/__init__.py:30)" % net_type)


class LinLayers(nn.ModuleList):
    def __init__(self, n_channels_list: Sequence[int]):
        super().__init__([nn.Sequential(nn.Identity(), nn.Conv2d(nc, 1, 1, 1, 0, bias=False)) for nc in n_channels_list])
        for param in self.parameters():
            param.requires_grad = False


def alexnet_features():
    """torchvision.models.alexnet().features"""
    return nn.Sequential(
        nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
        nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
        nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2))


class BaseNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("mean", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("std", torch.Tensor([.458, .448, .450])[None, :, None, None])

    def set_requires_grad(self, state: bool):
        for param in chain(self.parameters(), self.buffers()):
            param.requires_grad = state

    def nhwc(self, x):
        """x NHWC -> list of unit-normalised NHWC feature maps at the target layers (networks.py:52-62)"""
        with weight_scope(self):
            n, h, w, c = x.shape
            # z_score through the resize kernel's (v - mean) / std epilogue at identical size
            x = FF.resize_bilinear(x, h, w, self.mean.view(-1).contiguous(), self.std.view(-1).contiguous())
            output = []
            for i, layer in enumerate(self.layers, 1):
                if isinstance(layer, nn.Conv2d):
                    x = run_conv(layer, x)
                elif isinstance(layer, nn.ReLU):
                    x = FF.leaky_relu(x, 0.0)
                elif isinstance(layer, nn.MaxPool2d):
                    x = FF.max_pool(x, layer.kernel_size, layer.stride)
                else:
                    raise NotImplementedError(type(layer))
                if i in self.target_layers:
                    output.append(normalize_activation(x))
                if len(output) == len(self.target_layers):
                    break
            return output

    def forward(self, x):
        return [FF.to_nchw(f) for f in self.nhwc(FF.to_nhwc(x))]


class AlexNet(BaseNet):
    def __init__(self):
        super().__init__()
        self.layers = alexnet_features()
        self.target_layers = [2, 5, 8, 10, 12]
        self.n_channels_list = [64, 192, 384, 256, 256]
        self.set_requires_grad(False)
