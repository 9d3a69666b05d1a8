"""Host-side mirror of modules/example_guided_att.py (example_guided_att.py:5-41): same constructor, parameter
names (``conv.weight``, ``out_conv.weight/bias``) and forward(src_mask, src_feature, ref_feature).  The
[N, HW, HW] attention map is produced chunk-wise inside the Infinity Cache and applied to BOTH value maps from
one softmax pass; the mask blend and the channel concat are one kernel."""
from __future__ import annotations

from torch import nn

from .. import functional as FF
from ..weights import weight_scope
from .pluralistic_model.external_function import run_conv


class ExampleGuidedAttention(nn.Module):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels // 4, 1, bias=False)
        self.out_channels = out_channels
        if out_channels is not None:
            self.out_conv = nn.Conv2d(in_channels * 2, out_channels, 1)

    def nhwc(self, mask_nhw, src, ref):
        """mask [N,H,W] (soft), src / ref [N,H,W,C] -> [N,H,W,2C] (or out_channels)."""
        with weight_scope(self):
            n, h, w, c = src.shape
            q = run_conv(self.conv, src)
            src_att, ref_att = FF.self_attention(q.view(n, h * w, -1), [src.view(n, h * w, c), ref.view(n, h * w, c)])
            out = FF.guide_blend_cat(ref_att.view(n, h, w, c), ref, src_att.view(n, h, w, c), mask_nhw)
            if self.out_channels is not None:
                out = run_conv(self.out_conv, out)
            return out

    def forward(self, src_mask, src_feature, ref_feature):
        """src_mask [N,1,H,W]; src_feature, ref_feature [N,C,H,W] -> [N, 2C | out_channels, H, W]"""
        m = src_mask.reshape(src_mask.shape[0], src_mask.shape[2], src_mask.shape[3]).contiguous()
        return FF.to_nchw(self.nhwc(m, FF.to_nhwc(src_feature), FF.to_nhwc(ref_feature)))
