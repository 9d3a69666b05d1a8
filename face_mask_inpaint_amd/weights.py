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

import os
from typing import List

import torch
from torch import nn

from . import functional as FF

_ACTIVE = [0]
KEEP_FROZEN_PACKS = os.environ.get("FMI_FROZEN_PACKS", "1") != "0"  # off: every weight is re-packed every forward (A/B)


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
                items, todo = [], []
                for c in convs:
                    w3 = not getattr(c, "_fmi_no_w3", False)  # set on convolutions that run on bf16 activations
                    if hasattr(c, "weight_bar"):
                        items.append((c.weight_bar, c.weight_u, c.weight_v, w3, getattr(c, "_fmi_power_iterations", 1)))
                        todo.append((c, None))
                        continue
                    # A FROZEN weight (the loss networks of train_psp.py: ArcFace IR-SE50, VGG19, AlexNet; every network at inference) is
                    # packed once and the packs are kept while the tensor is unchanged -- same storage, same version counter, same
                    # want_w3.  Trainable weights are re-packed every forward: the fused optimiser writes them through raw pointers,
                    # which the version counter does not see.  (In-place edits through ``.data`` bypass the counter as well:
                    # ``invalidate_packs(module)`` after such an edit.)
                    w = c.weight
                    key = None if (w.requires_grad or not KEEP_FROZEN_PACKS) else (w.data_ptr(), w._version, tuple(w.shape), w3)
                    cached = getattr(c, "_fmi_frozen_pack", None)
                    if key is not None and cached is not None and cached[0] == key:
                        object.__setattr__(c, "_fmi_packed", cached[1])
                        continue
                    items.append((w, None, None, w3))
                    todo.append((c, key))
                if items:
                    for (c, key), pw in zip(todo, FF.prepare_weights(items)):
                        object.__setattr__(c, "_fmi_packed", pw)
                        object.__setattr__(c, "_fmi_frozen_pack", None if key is None else (key, pw))
        _ACTIVE[0] += 1
        return self

    def __exit__(self, *exc):
        _ACTIVE[0] -= 1
        return False


def invalidate_packs(root: nn.Module) -> None:
    """forget the kept packs of frozen weights under ``root`` (needed only after edits that bypass the tensor's version counter)"""
    for m in root.modules():
        if hasattr(m, "_fmi_frozen_pack"):
            object.__setattr__(m, "_fmi_frozen_pack", None)


def packed(conv: nn.Module) -> FF.PackedWeight:
    pw = getattr(conv, "_fmi_packed", None)
    if pw is None:
        raise FF.FmiError("conv weights were not prepared: call the block inside weight_scope")
    return pw
