"""Per-forward weight preparation scope.

The reference re-normalises every SpectralNorm-wrapped conv inside that conv's own forward
(external_function.py:70-72: ~6 tiny kernels per conv).  Here the outermost module of a forward pass
collects every conv that will run and prepares them in ONE launch (spectral norm + packing into the layouts the
implicit-GEMM kernels read); nested blocks find their packed weights already in place.  Semantics are unchanged:
each conv's u/v advance exactly once per top-level forward, as in the reference, because every conv runs once
per forward.  Sub-trees that the reference never executes (Auto_Attn.model when ``pre is None``,
base_function.py:441-446) are excluded so that their u/v stay untouched.
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from . import functional as FF

_ACTIVE = [0]


def _collect(root: nn.Module, skip=()) -> List[nn.Module]:
    key = tuple(sorted(id(m) for m in skip))
    cache = getattr(root, "_fmi_weight_lists", None)
    if cache is None:
        cache = {}
        object.__setattr__(root, "_fmi_weight_lists", cache)
    if key in cache:
        return cache[key]
    out, seen = [], set()
    skip_ids = set(key)

    def walk(m: nn.Module):
        if getattr(m, "_fmi_never_runs", False) or id(m) in skip_ids:
            return
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if id(m) not in seen:
                seen.add(id(m))
                out.append(m)
            return
        for ch in m.children():
            walk(ch)

    walk(root)
    cache[key] = out
    return out


class weight_scope:
    """``with weight_scope(module):`` -- prepares all conv weights under ``module`` unless an outer scope did.  ``skip``: sub-modules
    that will NOT run in this forward (e.g. the decoder's latent ResBlocks when ReferenceFill is called with no_prior): their
    SpectralNorm u / v must stay untouched, as in the reference where a conv's power iteration happens inside its own forward."""

    def __init__(self, root: nn.Module, skip=()):
        self.root = root
        self.skip = tuple(skip)
        self.owner = False

    def __enter__(self):
        if _ACTIVE[0] == 0:
            self.owner = True
            convs = _collect(self.root, self.skip)
            if convs:
                items = []
                for c in convs:
                    w3 = not getattr(c, "_fmi_no_w3", False)  # set on convolutions that run on bf16 activations
                    if hasattr(c, "weight_bar"):
                        items.append((c.weight_bar, c.weight_u, c.weight_v, w3, getattr(c, "_fmi_power_iterations", 1)))
                    else:
                        items.append((c.weight, None, None, w3))
                for c, pw in zip(convs, FF.prepare_weights(items)):
                    object.__setattr__(c, "_fmi_packed", pw)
        _ACTIVE[0] += 1
        return self

    def __exit__(self, *exc):
        _ACTIVE[0] -= 1
        return False


def packed(conv: nn.Module) -> FF.PackedWeight:
    pw = getattr(conv, "_fmi_packed", None)
    if pw is None:
        raise FF.FmiError("conv weights were not prepared: call the block inside weight_scope")
    return pw
