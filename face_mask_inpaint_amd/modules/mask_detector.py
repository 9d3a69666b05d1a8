"""Host-side mirror of modules/mask_detector.py:8-30: UNet(3 -> 2) segmentation of the face mask.  ``forward(image, 'train')``
returns the logits [N, 2, H, W]; any other mode returns ``softmax(logits) > threshold`` as the reference does.  ``predict_mask``
is the call PICNet_inference.infer_batch makes (``.argmax(1).float()``, PICNet_inference.py:100-101) as one index kernel."""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as FF
from .unet.unet_model import UNet


class MaskDetector(nn.Module):
    def __init__(self, n_channels, bilinear=True, threshold=0.5):
        super().__init__()
        self.model = UNet(n_channels, 2, bilinear=bilinear)
        self.threshold = threshold
        self.n_channels = n_channels
        self.bilinear = bilinear
        self.n_classes = 2

    def forward(self, image, mode="train"):
        output = self.model(image)
        if mode == "train":
            return output
        # two classes: softmax(l)_c > t  <=>  sigmoid(l_c - l_other) > t; evaluated on [N, 2, H, W] like the reference
        probs = torch.softmax(output, dim=1)
        return probs > self.threshold

    @torch.no_grad()
    def predict_mask(self, image):
        """[N, H, W] float {0, 1} = forward(image, 'train').argmax(1).float() -- logits stay NHWC, the argmax is bit exact"""
        return FF.argmax_channels(self.model.nhwc(FF.to_nhwc(image)))
