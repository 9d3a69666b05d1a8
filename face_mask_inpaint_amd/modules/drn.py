"""Host-side mirror of modules/drn.py: the dilated residual networks DRN-C / DRN-D (BasicBlock, Bottleneck, DRN and the
``drn_c_* / drn_d_*`` constructors) -- the ``--encoder_type drn`` alternative of ReferenceFill (model.py:47-59 builds two
``drn_c_42(pretrained=False, out_map=True)`` and replaces ``fc`` by a 1x1 conv to ``img_f`` channels).  Same parameter names
(``layerN.M.conv1/bn1/...``, ``downsample.0/1``, ``fc``), same initialisation; forward on the HIP kernels: dilation is the tap step
of the implicit-GEMM gather (``fmi_conv_desc.dil``), BatchNorm follows ``self.training``.  ``pretrained=True`` needs a download
(model_zoo URLs, drn.py:16-24) and raises; ``DRN_A`` / ``drn_a_50`` (a torchvision-ResNet variant nobody constructs) is absent."""
from __future__ import annotations

import math

import torch.nn as nn

from .. import functional as FF
from ..weights import weight_scope
from .pluralistic_model.external_function import run_conv
from .psp.encoders.helpers import batch_norm

BatchNorm = nn.BatchNorm2d


def conv3x3(in_planes, out_planes, stride=1, padding=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=padding, bias=False, dilation=dilation)


def _relu(x):
    return FF.leaky_relu(x, 0.0)


class _Block(nn.Module):
    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))

    def _residual(self, x):
        if self.downsample is None:
            return x
        return batch_norm(self.downsample[1], run_conv(self.downsample[0], x))


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=(1, 1), residual=True):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride, padding=dilation[0], dilation=dilation[0])
        self.bn1 = BatchNorm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, padding=dilation[1], dilation=dilation[1])
        self.bn2 = BatchNorm(planes)
        self.downsample = downsample
        self.stride = stride
        self.residual = residual

    def nhwc(self, x):
        with weight_scope(self):
            out = _relu(batch_norm(self.bn1, run_conv(self.conv1, x)))
            out = batch_norm(self.bn2, run_conv(self.conv2, out))
            if self.residual:
                out = FF.add(out, self._residual(x))
            return _relu(out)


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=(1, 1), residual=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation[1], bias=False, dilation=dilation[1])
        self.bn2 = BatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def nhwc(self, x):
        with weight_scope(self):
            out = _relu(batch_norm(self.bn1, run_conv(self.conv1, x)))
            out = _relu(batch_norm(self.bn2, run_conv(self.conv2, out)))
            out = batch_norm(self.bn3, run_conv(self.conv3, out))
            return _relu(FF.add(out, self._residual(x)))


class DRN(nn.Module):
    def __init__(self, block, layers, num_classes=1000, channels=(16, 32, 64, 128, 256, 512, 512, 512), out_map=False, out_middle=False,
                 pool_size=28, arch="D"):
        super().__init__()
        self.inplanes = channels[0]
        self.out_map = out_map
        self.out_dim = channels[-1]
        self.out_middle = out_middle
        self.arch = arch
        if arch == "C":
            self.conv1 = nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False)
            self.bn1 = BatchNorm(channels[0])
            self.relu = nn.ReLU(inplace=True)
            self.layer1 = self._make_layer(BasicBlock, channels[0], layers[0], stride=1)
            self.layer2 = self._make_layer(BasicBlock, channels[1], layers[1], stride=2)
        elif arch == "D":
            self.layer0 = nn.Sequential(nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False), BatchNorm(channels[0]),
                                        nn.ReLU(inplace=True))
            self.layer1 = self._make_conv_layers(channels[0], layers[0], stride=1)
            self.layer2 = self._make_conv_layers(channels[1], layers[1], stride=2)
        self.layer3 = self._make_layer(block, channels[2], layers[2], stride=2)
        self.layer4 = self._make_layer(block, channels[3], layers[3], stride=2)
        self.layer5 = self._make_layer(block, channels[4], layers[4], dilation=2, new_level=False)
        self.layer6 = None if layers[5] == 0 else self._make_layer(block, channels[5], layers[5], dilation=4, new_level=False)
        if arch == "C":
            self.layer7 = None if layers[6] == 0 else self._make_layer(BasicBlock, channels[6], layers[6], dilation=2, new_level=False, residual=False)
            self.layer8 = None if layers[7] == 0 else self._make_layer(BasicBlock, channels[7], layers[7], dilation=1, new_level=False, residual=False)
        elif arch == "D":
            self.layer7 = None if layers[6] == 0 else self._make_conv_layers(channels[6], layers[6], dilation=2)
            self.layer8 = None if layers[7] == 0 else self._make_conv_layers(channels[7], layers[7], dilation=1)
        if num_classes > 0:
            self.avgpool = nn.AvgPool2d(pool_size)
            self.fc = nn.Conv2d(self.out_dim, num_classes, kernel_size=1, stride=1, padding=0, bias=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, BatchNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, new_level=True, residual=True):
        assert dilation == 1 or dilation % 2 == 0
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       BatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample,
                        dilation=(1, 1) if dilation == 1 else (dilation // 2 if new_level else dilation, dilation), residual=residual)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, residual=residual, dilation=(dilation, dilation)))
        return nn.Sequential(*layers)

    def _make_conv_layers(self, channels, convs, stride=1, dilation=1):
        modules = []
        for i in range(convs):
            modules.extend([nn.Conv2d(self.inplanes, channels, kernel_size=3, stride=stride if i == 0 else 1, padding=dilation, bias=False,
                                      dilation=dilation), BatchNorm(channels), nn.ReLU(inplace=True)])
            self.inplanes = channels
        return nn.Sequential(*modules)

    @staticmethod
    def _run(seq, x):
        """a layer: nn.Sequential of blocks (arch C, layers 3-6) or of Conv2d / BatchNorm / ReLU triples (arch D)"""
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _Block):
                x = m.nhwc(x)
                i += 1
            else:  # Conv2d, BatchNorm, ReLU
                x = _relu(batch_norm(mods[i + 1], run_conv(m, x)))
                i += 3
        return x

    def nhwc(self, x):
        with weight_scope(self):
            y = []
            if self.arch == "C":
                x = _relu(batch_norm(self.bn1, run_conv(self.conv1, x)))
            else:
                x = self._run(self.layer0, x)
            for name in ("layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "layer7", "layer8"):
                layer = getattr(self, name)
                if layer is not None:
                    x = self._run(layer, x)
                    y.append(x)
            if self.out_map:
                x = run_conv(self.fc, x)
            else:
                k = self.avgpool.kernel_size
                x = run_conv(self.fc, FF.avg_pool(x, k if isinstance(k, int) else k[0]))
                x = x.reshape(x.shape[0], -1)
            return x, y

    def forward(self, x):
        out, y = self.nhwc(FF.to_nhwc(x))
        out = FF.to_nchw(out) if out.ndim == 4 else out
        if self.out_middle:
            return out, [FF.to_nchw(t) for t in y]
        return out


def _no_download(pretrained):
    if pretrained:
        raise NotImplementedError("pretrained DRN weights are a network download (drn.py:16-24); load a local state_dict instead")


def drn_a_50(pretrained=False, **kwargs):
    raise NotImplementedError("DRN_A (drn.py:262-330) is not built: no caller of the reference constructs it")


def drn_c_26(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 2, 2, 2, 2, 1, 1], arch="C", **kwargs)


def drn_c_42(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 3, 4, 6, 3, 1, 1], arch="C", **kwargs)


def drn_c_58(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(Bottleneck, [1, 1, 3, 4, 6, 3, 1, 1], arch="C", **kwargs)


def drn_d_22(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 2, 2, 2, 2, 1, 1], arch="D", **kwargs)


def drn_d_24(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 2, 2, 2, 2, 2, 2], arch="D", **kwargs)


def drn_d_38(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 3, 4, 6, 3, 1, 1], arch="D", **kwargs)


def drn_d_40(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(BasicBlock, [1, 1, 3, 4, 6, 3, 2, 2], arch="D", **kwargs)


def drn_d_54(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(Bottleneck, [1, 1, 3, 4, 6, 3, 1, 1], arch="D", **kwargs)


def drn_d_56(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(Bottleneck, [1, 1, 3, 4, 6, 3, 2, 2], arch="D", **kwargs)


def drn_d_105(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(Bottleneck, [1, 1, 3, 4, 23, 3, 1, 1], arch="D", **kwargs)


def drn_d_107(pretrained=False, **kwargs):
    _no_download(pretrained)
    return DRN(Bottleneck, [1, 1, 3, 4, 23, 3, 2, 2], arch="D", **kwargs)
