"""Host-side mirror of modules/psp/encoders/helpers.py (get_blocks, SEModule, bottleneck_IR, bottleneck_IR_SE) with the
reference's parameter names (``res_layer.N.*``, ``shortcut_layer.N.*``, ``fc1/fc2``); forward runs on the HIP kernels in
NHWC.  BatchNorm2d uses batch statistics in training mode (running statistics are updated like torch does) and the
running statistics in eval mode."""
from __future__ import annotations

from collections import namedtuple

import torch
from torch.nn import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, MaxPool2d, Module, PReLU, ReLU, Sequential, Sigmoid

from .... import functional as FF
from ....weights import weight_scope
from ...pluralistic_model.external_function import run_conv


class Flatten(Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


class Bottleneck(namedtuple("Block", ["in_channel", "depth", "stride"])):
    """A named tuple describing a ResNet block."""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


def get_blocks(num_layers):
    units = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}
    if num_layers not in units:
        raise ValueError("Invalid number of layers: {}. Must be one of [50, 100, 152]".format(num_layers))
    u = units[num_layers]
    return [get_block(64, 64, u[0]), get_block(64, 128, u[1]), get_block(128, 256, u[2]), get_block(256, 512, u[3])]


# > 1 while a module runs a batch made of that many parts that are SEPARATE forward calls in the reference (GradualStyleEncoder runs
# its body on the source and then on the reference image): training-mode BatchNorm then normalises and updates per part, in order
BN_GROUPS = [1]
_FUSE = __import__("os").environ.get("FMI_IRSE_FUSE_OFF") is None  # debug: A/B of the block-level fusions below
_GATE_PASS = __import__("os").environ.get("FMI_IRSE_GATE_PASS", "1") != "0"  # debug: A/B of the SE input's one-pass gradient join (fp32)


def batch_norm(bn: BatchNorm2d, x, passthrough=False):
    """nn.BatchNorm2d forward on NHWC with torch's training / eval semantics.  ``passthrough``: returns (y, x') where x' is x for its
    other consumer; in training mode that consumer's gradient is added inside the BatchNorm backward kernel (one pass fewer)"""
    if bn.training or not bn.track_running_stats:
        groups = BN_GROUPS[0]
        if passthrough:
            y, stats, sums, xp = FF.batch_norm_train(x, bn.weight, bn.bias, bn.eps, groups, True)
        else:
            y, stats, sums = FF.batch_norm_train(x, bn.weight, bn.bias, bn.eps, groups)
            xp = None
        y = (y, xp) if passthrough else y
        if bn.training and bn.track_running_stats:
            cnt = x.numel() // x.shape[-1] // groups
            for g in range(groups):  # the running statistics see the parts one after the other, as the reference's separate calls do
                st, sm = stats[g:g + 1], sums[g:g + 1]
                if bn.momentum is not None and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous():
                    with torch.no_grad():  # one launch: momentum update of both buffers and the batch counter
                        FF.batch_norm_running_update(st, bn.running_mean, bn.running_var, bn.num_batches_tracked, cnt, bn.eps, bn.momentum, sm)
                    continue
                with torch.no_grad():  # cumulative moving average (momentum=None): [C]-sized torch bookkeeping
                    mean = st[0, :, 0]
                    m64 = sm[0, :, 0] / cnt
                    var = ((sm[0, :, 1] / cnt - m64 * m64).clamp_min(0) * (cnt / max(cnt - 1, 1))).float()
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var, alpha=m)
                    bn.num_batches_tracked += 1
        return y
    if x.dtype == torch.bfloat16:  # eval-mode BatchNorm of a bf16 body: evaluated in fp32 (inference path, not the benchmarked one)
        y = batch_norm(bn, x.float()).to(torch.bfloat16)
        return (y, x) if passthrough else y
    # eval mode: frozen statistics, but the affine parameters still receive gradients like torch.nn.BatchNorm2d's do
    # ([C]-sized torch bookkeeping; the per-pixel work and its reductions are the library's)
    want = torch.is_grad_enabled() and (bn.weight.requires_grad or bn.bias.requires_grad)
    if not want:
        # frozen statistics AND frozen affine parameters (ArcFace inside IDLoss: ~50 BatchNorms, two forwards per step): scale / shift
        # are constants -- five [C]-sized launches per BatchNorm and forward otherwise.  Kept while none of the four tensors changes
        # (storage + version counter, as weights.weight_scope keeps the packs of frozen convolution weights).
        key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_var, bn.running_mean)) + (bn.eps,)
        kept = getattr(bn, "_fmi_affine", None)
        if kept is not None and kept[0] == key:
            scale, shift = kept[1], kept[2]
        else:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                shift = bn.bias - bn.running_mean * scale
            object.__setattr__(bn, "_fmi_affine", (key, scale, shift))
    else:
        with torch.enable_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
    if passthrough and _GATE_PASS and __import__("os").environ.get("FMI_IRSE_GATE_PASS", "1") != "2":  # frozen network: the other consumer's gradient joins inside this op's backward pass
        yp = FF.channel_affine(x, scale, shift, passthrough=True)
        if yp is not None:
            return yp
    y = FF.channel_affine(x, scale, shift)
    return (y, x) if passthrough else y


class SEModule(Module):
    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def gate(self, x):
        """the [N, C] channel gates sigmoid(fc2(relu(fc1(avgpool(x))))) (fp32)"""
        n, h, w, c = x.shape
        s = FF.avg_pool(x, h) if h == w else FF.adaptive_avg_pool(x, 1, 1)  # AdaptiveAvgPool2d(1) -> [N,1,1,C]
        s = FF.leaky_relu(run_conv(self.fc1, s), 0.0)
        return FF.sigmoid(run_conv(self.fc2, s)).view(n, c)

    def gate_pass(self, x):
        """fp32 activations: (gates [N, C], x') -- x' is x handed on to the gated product, so that x's two gradients meet in one kernel
        instead of a pool-backward pass plus an accumulation pass (functional._GlobalAvgPoolPassF32); None where that op does not apply"""
        n, h, w, c = x.shape
        if not _GATE_PASS or h != w or c % 4 != 0 or x.dtype != torch.float32:
            return None
        pooled, xp = FF.global_avg_pool_pass(x)
        s = FF.leaky_relu(run_conv(self.fc1, pooled), 0.0)
        return FF.sigmoid(run_conv(self.fc2, s)).view(n, c), xp

    def gate_bf16(self, x):
        """bf16 activations: (gates fp32 [N, C], x') -- x' is x handed on to the gated product so that both gradients of x meet in
        one kernel (functional._GlobalAvgPoolPassBF16); the two tiny fully-connected layers stay fp32"""
        n, h, w, c = x.shape
        pooled, xp = FF.global_avg_pool_pass_bf16(x)
        s = FF.leaky_relu(run_conv(self.fc1, pooled.view(n, 1, 1, c)), 0.0)
        return FF.sigmoid(run_conv(self.fc2, s)).view(n, c), xp

    def nhwc(self, x):
        with weight_scope(self):
            return FF.scale_channels(x, self.gate(x))

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))


class _Bottleneck(Module):
    def nhwc(self, x):
        with weight_scope(self):
            if _FUSE and isinstance(self.shortcut_layer, MaxPool2d) and self._stride == 1:
                # identity shortcut: x has two consumers (BatchNorm and the final add); the add's gradient re-enters through the
                # BatchNorm backward kernel instead of an accumulation pass of its own
                r, sc = batch_norm(self.res_layer[0], x, passthrough=True)
            else:
                sc = FF.subsample(x, self._stride) if isinstance(self.shortcut_layer, MaxPool2d) else \
                    batch_norm(self.shortcut_layer[1], run_conv(self.shortcut_layer[0], x))
                r = batch_norm(self.res_layer[0], x)
            r = run_conv(self.res_layer[1], r)
            r = FF.prelu(r, self.res_layer[2].weight)
            r = run_conv(self.res_layer[3], r)
            r = batch_norm(self.res_layer[4], r)
            if len(self.res_layer) > 5 and r.dtype == torch.bfloat16:
                gates, rp = self.res_layer[5].gate_bf16(r)
                return FF.scale_channels_add(rp, gates, sc)
            if len(self.res_layer) > 5 and _FUSE:  # SE gate and residual add in one pass
                gp = self.res_layer[5].gate_pass(r)
                if gp is not None:
                    return FF.scale_channels_add(gp[1], gp[0], sc)
                return FF.scale_channels_add(r, self.res_layer[5].gate(r), sc)
            if len(self.res_layer) > 5:
                r = self.res_layer[5].nhwc(r)
            return FF.add(r, sc)

    def forward(self, x):
        return FF.to_nchw(self.nhwc(FF.to_nhwc(x)))


def _layers(in_channel, depth, stride, se):
    res = [BatchNorm2d(in_channel), Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
           Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth)]
    if se:
        res.append(SEModule(depth, 16))
    return Sequential(*res)


class bottleneck_IR(_Bottleneck):
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self._stride = stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        self.res_layer = _layers(in_channel, depth, stride, se=False)


class bottleneck_IR_SE(_Bottleneck):
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self._stride = stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        self.res_layer = _layers(in_channel, depth, stride, se=True)
