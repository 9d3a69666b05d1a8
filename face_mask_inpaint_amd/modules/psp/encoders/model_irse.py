"""Host-side mirror of modules/psp/encoders/model_irse.py: the ArcFace ``Backbone`` (IR / IR-SE 50 / 100 / 152) used by IDLoss,
with the reference's parameter names; forward on the HIP kernels.  Dropout follows ``self.training`` (IDLoss keeps it in eval)."""
from __future__ import annotations

import torch
from torch.nn import BatchNorm1d, BatchNorm2d, Conv2d, Dropout, Linear, Module, PReLU, Sequential

from .... import functional as FF
from ....weights import weight_scope
from ...pluralistic_model.external_function import run_conv
from .helpers import Flatten, batch_norm, bottleneck_IR, bottleneck_IR_SE, get_blocks


def l2_norm(input, axis=1):
    """helpers.py:15-18 for [N, C] embeddings"""
    if axis != 1 or input.ndim != 2:
        raise NotImplementedError("l2_norm: [N, C] embeddings, axis 1")
    return FF.l2norm_rows(input, 0.0)


class Backbone(Module):
    def __init__(self, input_size, num_layers, mode="ir", drop_ratio=0.4, affine=True):
        super().__init__()
        assert input_size in [112, 224], "input_size should be 112 or 224"
        assert num_layers in [50, 100, 152], "num_layers should be 50, 100 or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        unit_module = bottleneck_IR if mode == "ir" else bottleneck_IR_SE
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        side = 7 if input_size == 112 else 14
        self.output_layer = Sequential(BatchNorm2d(512), Dropout(drop_ratio), Flatten(), Linear(512 * side * side, 512),
                                       BatchNorm1d(512, affine=affine))
        self.body = Sequential(*[unit_module(b.in_channel, b.depth, b.stride) for block in get_blocks(num_layers) for b in block])

    def nhwc(self, x):
        with weight_scope(self):
            x = FF.prelu(batch_norm(self.input_layer[1], run_conv(self.input_layer[0], x)), self.input_layer[2].weight)
            for blk in self.body:
                x = blk.nhwc(x)
            x = batch_norm(self.output_layer[0], x)
            if self.training and self.output_layer[1].p > 0:
                raise NotImplementedError("Backbone in training mode (Dropout): IDLoss keeps the network in eval (id_loss.py:20)")
            n = x.shape[0]
            x = x.permute(0, 3, 1, 2).reshape(n, -1)  # nn.Flatten of the NCHW tensor: (c, h, w) order of the Linear's columns
            lin = self.output_layer[3]
            x = FF.linear(x, lin.weight, lin.bias)
            bn = self.output_layer[4]
            if bn.affine:
                x = batch_norm(bn, x.view(n, 1, 1, -1)).view(n, -1)
            else:
                if bn.training:
                    raise NotImplementedError("BatchNorm1d(affine=False) in training mode")
                with torch.no_grad():
                    scale = torch.rsqrt(bn.running_var + bn.eps)
                    shift = -bn.running_mean * scale
                x = FF.channel_affine(x.view(n, 1, 1, -1), scale, shift).view(n, -1)
            return l2_norm(x)

    def forward(self, x):
        return self.nhwc(FF.to_nhwc(x))


def IR_50(input_size):
    return Backbone(input_size, num_layers=50, mode="ir", drop_ratio=0.4, affine=False)


def IR_101(input_size):
    return Backbone(input_size, num_layers=100, mode="ir", drop_ratio=0.4, affine=False)


def IR_152(input_size):
    return Backbone(input_size, num_layers=152, mode="ir", drop_ratio=0.4, affine=False)


def IR_SE_50(input_size):
    return Backbone(input_size, num_layers=50, mode="ir_se", drop_ratio=0.4, affine=False)


def IR_SE_101(input_size):
    return Backbone(input_size, num_layers=100, mode="ir_se", drop_ratio=0.4, affine=False)


def IR_SE_152(input_size):
    return Backbone(input_size, num_layers=152, mode="ir_se", drop_ratio=0.4, affine=False)
