"""Host-side mirror of modules/psp/criteria/lpips/utils.py: normalize_activation (:6-8) on the HIP kernel, and get_state_dict
(:11-30) -- the reference downloads the linear heads from a URL; there is no network here, so a LOCAL file is read instead
(``FMI_LPIPS_LIN_<NET>`` or ``pretrained_models/lpips_<net>.pth``, tensor-only, ``weights_only=True``) and None is returned when
it is absent (the caller keeps its random initialisation and says so)."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from ..... import functional as FF


def normalize_activation(x_nhwc, eps=1e-10):
    """x / (sqrt(sum_c x^2) + eps) per pixel; NHWC in, NHWC out"""
    return FF.l2norm_rows(x_nhwc, eps)


def get_state_dict(net_type: str = "alex", version: str = "0.1"):
    path = os.environ.get("FMI_LPIPS_LIN_" + net_type.upper(), os.path.join("pretrained_models", f"lpips_{net_type}.pth"))
    if not os.path.isfile(path):
        return None
    old_state_dict = torch.load(path, map_location="cpu", weights_only=True)
    new_state_dict = OrderedDict()
    for key, val in old_state_dict.items():  # the renaming of utils.py:22-28
        new_state_dict[key.replace("lin", "").replace("model.", "")] = val
    return new_state_dict
