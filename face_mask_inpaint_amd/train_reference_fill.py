"""Host-side mirror of the model-side helpers of train_reference_fill.py: ``process_params`` (:88-104) and ``load_networks``
(:107-140, the shape-matched partial load of PICNet checkpoints ``latest_net_{G,E,D}.pth``).

``load_networks`` reproduces the reference literally -- including that for G and E it collects the MODEL'S OWN tensors for
every key whose shape matches the checkpoint (``matches[k] = v`` at :123-126 takes ``v`` from ``generator.decoder.state_dict()``,
not from the file), so those two loads are value-preserving, and a key missing from the file raises KeyError exactly as there;
only D is really loaded (strict).  ``copy_pretrained=True`` is this build's fix: the matching tensors are taken from the file."""
from __future__ import annotations

import os

import torch


def process_params(args):
    encoder_params = {k.replace("encoder_", ""): v for k, v in args._get_kwargs() if k.startswith("encoder")}
    decoder_params = {k.replace("decoder_", ""): v for k, v in args._get_kwargs() if k.startswith("decoder")}
    disc_params = {k.replace("disc_", ""): v for k, v in args._get_kwargs() if k.startswith("disc")}
    disc_params["img_f"] = encoder_params["img_f"]
    return encoder_params, decoder_params, disc_params


def _matches(module, pretrained_dict, copy_pretrained):
    out = {}
    for k, v in module.state_dict().items():
        if v.shape == pretrained_dict[k].shape:  # KeyError for a key the checkpoint lacks, as in the reference
            out[k] = pretrained_dict[k] if copy_pretrained else v
    return out


def load_networks(generator, discriminator, path, copy_pretrained=False):
    """Load all the networks from the disk (train_reference_fill.py:107-140).  Checkpoints are read with
    ``weights_only=True`` (tensor-only state_dicts; nothing from the file is executed)."""
    if not path:
        return
    for name in ["G", "E", "D"]:
        ckpt_path = os.path.join(path, f"latest_net_{name}.pth")
        if not os.path.isfile(ckpt_path):
            continue
        pretrained_dict = {k.replace("module.", "", 1): v for k, v in torch.load(ckpt_path, map_location="cpu", weights_only=True).items()}
        if name == "G":
            generator.decoder.load_state_dict(_matches(generator.decoder, pretrained_dict, copy_pretrained), strict=False)
        elif name == "E":
            generator.src_encoder.load_state_dict(_matches(generator.src_encoder, pretrained_dict, copy_pretrained), strict=False)
            generator.ref_encoder.load_state_dict(_matches(generator.ref_encoder, pretrained_dict, copy_pretrained), strict=False)
        elif name == "D":
            discriminator.load_state_dict(pretrained_dict, strict=True)  # discriminator did not change: strict loading
