"""Host-side mirror of modules/psp/criteria/lpips/lpips.py: LPIPS(net_type='alex').  The reference pins its tensors to "cuda"
(:24,27) and downloads both the AlexNet weights and the linear heads; here the modules follow ``.to(device)`` like any other and
the weights come from local files when present (utils.get_state_dict, ``FMI_LPIPS_ALEX`` for the trunk), random initialisation
otherwise -- WEIGHT parity is therefore unpinned, the arithmetic is pinned by tests/golden/psp_criteria.pt."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..... import functional as FF
from .networks import LinLayers, get_network
from .utils import get_state_dict


class LPIPS(nn.Module):
    def __init__(self, net_type: str = "alex", version: str = "0.1"):
        assert version in ["0.1"], "v0.1 is only supported now"
        super().__init__()
        self.net = get_network(net_type)
        self.lin = LinLayers(self.net.n_channels_list)
        sd = get_state_dict(net_type, version)
        trunk = os.environ.get("FMI_LPIPS_" + net_type.upper(), os.path.join("pretrained_models", f"{net_type}net_features.pth"))
        if sd is not None:
            self.lin.load_state_dict(sd)
        if os.path.isfile(trunk):
            self.net.layers.load_state_dict(torch.load(trunk, map_location="cpu", weights_only=True))
        if sd is None or not os.path.isfile(trunk):
            print("LPIPS: pretrained weight files not found -> keeping the random initialisation")

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        n = x.shape[0]
        # both images through the trunk as ONE batch of 2N (same weights, half the launches)
        feats = self.net.nhwc(torch.cat([FF.to_nhwc(x), FF.to_nhwc(y)], dim=0))
        total = None
        for f, lin in zip(feats, self.lin):
            hw = f.shape[1] * f.shape[2]
            # (fx - fy)^2 -> 1x1 lin conv -> mean over (H, W) -> sum over the batch, / N   (lpips.py:33-36)
            v = FF.lpips_layer(f[:n], f[n:], lin[1].weight.view(-1), 1.0 / (hw * n))
            total = v if total is None else total + v
        return total
