"""Data path of the inpainting trainers as a host-index + device-preprocessing pipeline.

What the reference's ``dataloader.py`` offers (``BasicDataset`` :49-119, ``ReferenceDataset`` :122-266, ``get_reference_dataloader``
:19-47) is kept as an interface -- constructor arguments, the ``ids`` / ``identity_map`` / ``img2identity`` attributes, the item
dictionary ``src_img, gt_img, raw_gt_img, ref_img`` (float32 [3, H, W] in [0, 1], or [-1, 1] with ``apply_transform``), ``mask``
(int64 [H, W]) and optional ``id`` -- but the work is divided differently:

  host    directory listing, the identity file, jpg / npy decoding (PIL, numpy), Pillow's O(W + H) resampling tables;
  device  BICUBIC / NEAREST resizing in Pillow's integer arithmetic, uint8 HWC -> float32 CHW / 255, Normalize, mask binarisation and
          the SSIM scoring behind ``use_ssim`` -- kernels of libfmi_hip.so (``preprocess.DevicePreprocessor``, csrc/preproc.hip).

Items and batches therefore arrive as DEVICE tensors, equal bit for bit to what the reference's CPU pipeline returns for the same files
(tests/test_gpu_data.py against tests/golden/dataset.pt).  Batches are assembled by ``DeviceLoader``: files of a batch are decoded on a
small thread pool and preprocessed in one launch per tensor kind -- worker processes cannot hand GPU tensors around.  There is no CPU
preprocessing path.  The best-reference cache is a JSON file (the reference pickles it; foreign pickles are never loaded here).
"""
from __future__ import annotations

import json
import logging
import math
import os
import random
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .preprocess import DevicePreprocessor

_IMAGE_KEYS = ("src_img", "gt_img", "raw_gt_img", "ref_img")


def _stem_id(filename: str) -> str:
    """'000123_surgical.jpg' -> '000123': the part of the file name before the first underscore / the extension"""
    return os.path.splitext(filename)[0].split("_")[0]


def decode(path) -> np.ndarray:
    """file -> uint8 array ([H][W][3] for jpg / png, [H][W] for binary maps).  Only loaders that execute nothing from the file."""
    path = str(path)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        arr = np.load(path)  # allow_pickle stays False
    elif ext == ".npz":
        with np.load(path) as z:
            arr = z[z.files[0]]
    elif ext in (".pt", ".pth"):
        arr = torch.load(path, weights_only=True).numpy()
    else:
        from PIL import Image

        with Image.open(path) as im:
            arr = np.asarray(im)
    if arr.dtype != np.uint8:
        raise TypeError(f"{path}: expected 8-bit data, got {arr.dtype}")
    return np.ascontiguousarray(arr)


def parse_identity_file(path):
    """lines '<image file> <identity>' -> (identity -> [image ids in file order], image id -> identity)"""
    groups: Dict[int, List[str]] = {}
    owner: Dict[str, int] = {}
    with open(path, "r") as fh:
        for line in fh:
            fields = line.split()
            if len(fields) != 2:
                continue
            key, who = _stem_id(fields[0]), int(fields[1])
            owner[key] = who
            groups.setdefault(who, []).append(key)
    return groups, owner


class BasicDataset(torch.utils.data.Dataset):
    """``<id>_surgical.jpg`` images with ``<id><mask_suffix>.npy`` binary maps -> {'image', 'mask'} on the device."""

    def __init__(self, images_dir, masks_dir, scale=1.0, mask_suffix="", device=None):
        if not 0 < scale <= 1:
            raise AssertionError("Scale must be between 0 and 1")
        self.images_dir, self.masks_dir = str(images_dir), str(masks_dir)
        self.scale, self.mask_suffix = scale, mask_suffix
        self.ids = [_stem_id(f) for f in os.listdir(self.images_dir) if not f.startswith(".")]
        if not self.ids:
            raise RuntimeError(f"No input file found in {images_dir}, make sure you put your images there")
        self._pre: Optional[DevicePreprocessor] = None
        self._device = device
        logging.info("Creating dataset with %d examples", len(self.ids))

    @property
    def pre(self) -> DevicePreprocessor:
        if self._pre is None:
            self._pre = DevicePreprocessor(self._device)
        return self._pre

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, idx):
        key = self.ids[idx]
        img = decode(os.path.join(self.images_dir, key + "_surgical.jpg"))
        m = decode(os.path.join(self.masks_dir, key + self.mask_suffix + ".npy"))
        if img.shape[:2] != m.shape[:2]:
            raise AssertionError(f"Image and mask {key} should be the same size, but are {img.shape[:2]} and {m.shape[:2]}")
        return {"image": self.pre.images([img], self.scale)[0], "mask": self.pre.masks([m], self.scale)[0]}


class ReferenceDataset(BasicDataset):
    """source (masked) image, ground truth, a second image of the same identity and the binary map of the mask."""

    def __init__(self, source_dir, reference_dir, masks_dir, identity_file, apply_transform=True, scale=1.0, use_ssim=False, device=None,
                 return_id=False):
        if not 0 < scale <= 1:
            raise AssertionError("Scale must be between 0 and 1")
        self.source_dir, self.reference_dir, self.masks_dir = str(source_dir), str(reference_dir), str(masks_dir)
        self.scale, self.apply_transform, self.return_id, self.use_ssim = scale, apply_transform, return_id, use_ssim
        self.identity_map, self.img2identity = parse_identity_file(identity_file)
        # an identity with a single image has no reference partner: its image is left out
        self.filter_id = {members[0] for members in self.identity_map.values() if len(members) < 2}
        self.ids = [k for k in (_stem_id(f) for f in os.listdir(self.source_dir) if not f.startswith(".")) if k not in self.filter_id]
        if not self.ids:
            raise RuntimeError(f"No input file found in {source_dir}, make sure you put your images there")
        self._pre, self._device = None, device
        logging.info("Creating dataset with %d examples", len(self.ids))
        if use_ssim:
            cache = os.path.join(os.path.dirname(os.path.normpath(self.source_dir)), "best_reference_map.json")
            if os.path.isfile(cache):
                with open(cache) as fh:
                    self.best_reference_map = json.load(fh)
            else:
                self.best_reference_map = self.find_best_reference(device)
                with open(cache, "w") as fh:
                    json.dump(self.best_reference_map, fh)

    # ---- which second image ----
    def partners(self, key: str) -> List[str]:
        return [k for k in self.identity_map[self.img2identity[key]] if k != key]

    def find_best_reference(self, device=None) -> Dict[str, str]:
        """for every image the same-identity image with the highest SSIM to it, scored on the GPU by this library's SSIM kernel in one
        batch per image (dataloader.py:188-216 scores pair by pair with pytorch_msssim, which is absent offline: parity unpinned)"""
        from .modules.evaluations.ssim import ssim as ssim_fn

        if device is not None:
            self._device = device
        best = {}
        for key in self.ids:
            cands = self.partners(key)
            if not cands:
                best[key] = None
                continue
            imgs = self.pre.images([decode(self._ref_path(k)) for k in [key] + cands], self.scale)
            scores = [float(ssim_fn(imgs[:1], imgs[j + 1:j + 2])) for j in range(len(cands))]
            best[key] = cands[max(range(len(cands)), key=lambda j: (scores[j], -j))]  # first of equals, as a strict '>' scan keeps it
        return best

    def sample_reference_image(self, img_name: str) -> str:
        if self.use_ssim:
            return self.best_reference_map[img_name]
        pool = self.identity_map[self.img2identity[img_name]]
        if len(pool) < 2:
            raise AssertionError(f"identity of {img_name} has no second image")
        while True:  # draws from the WHOLE group and rejects the image itself: the random stream of the reference's sampler
            pick = random.choice(pool)
            if pick != img_name:
                return pick

    # ---- files ----
    def _ref_path(self, key):
        return os.path.join(self.reference_dir, key + ".jpg")

    def _files(self, key, partner):
        return (os.path.join(self.source_dir, key + "_surgical.jpg"), self._ref_path(key), self._ref_path(partner),
                os.path.join(self.masks_dir, key + ".npy"))

    def _load(self, key, partner):
        src, gt, ref, m = (decode(p) for p in self._files(key, partner))
        if src.shape[:2] != m.shape[:2]:
            raise AssertionError(f"Image and mask {key} should be the same size, but are {src.shape[:2]} and {m.shape[:2]}")
        return src, gt, ref, m

    # ---- tensors ----
    def assemble(self, keys: Sequence[str], partners: Sequence[str], decoded) -> Dict[str, torch.Tensor]:
        """decoded[i] = (src, gt, ref, mask) uint8 arrays of item i -> the batch dictionary on the device: one preprocessing launch
        sequence per tensor kind"""
        tr = bool(self.apply_transform)
        out = {"src_img": self.pre.images([d[0] for d in decoded], self.scale, normalise=tr),
               "ref_img": self.pre.images([d[2] for d in decoded], self.scale, normalise=tr)}
        out["gt_img"], out["raw_gt_img"] = self.pre.images([d[1] for d in decoded], self.scale, normalise=tr, also_plain=True)
        out["mask"] = self.pre.masks([d[3] for d in decoded], self.scale)
        if self.return_id:
            out["id"] = torch.tensor([[int(k)] for k in keys], dtype=torch.int64, device=out["mask"].device)
        return out

    def __getitem__(self, idx):
        key = self.ids[idx]
        partner = self.sample_reference_image(key)
        batch = self.assemble([key], [partner], [self._load(key, partner)])
        return {k: v[0] for k, v in batch.items()}


class DeviceLoader:
    """iterable over device batches of a ``ReferenceDataset`` subset (the role torch's DataLoader plays in the reference's trainers:
    ``len()`` = number of batches, ``.dataset`` / ``.indices``, a fresh shuffle per epoch)"""

    def __init__(self, dataset: ReferenceDataset, indices: Sequence[int], batch_size: int, shuffle=False, drop_last=False, num_workers=4):
        self.dataset, self.indices = dataset, list(indices)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.num_workers = max(1, int(num_workers))

    def __len__(self):
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        order = [self.indices[i] for i in torch.randperm(len(self.indices)).tolist()] if self.shuffle else list(self.indices)
        ds = self.dataset
        with ThreadPoolExecutor(self.num_workers) as pool:
            for b in range(len(self)):
                chunk = order[b * self.batch_size:(b + 1) * self.batch_size]
                keys = [ds.ids[i] for i in chunk]
                partners = [ds.sample_reference_image(k) for k in keys]  # on the calling thread: one random stream
                decoded = list(pool.map(lambda kp: ds._load(*kp), zip(keys, partners)))
                yield ds.assemble(keys, partners, decoded)


def get_reference_dataloader(dir_src_img, dir_ref_img, dir_mask, identity_file, batch_size, apply_transform=False, val_amount=0.1, num_workers=4,
                             img_scale=1.0, use_ssim=False, device=None):
    """(train loader, validation loader) over a random split; the training loader reshuffles every epoch, the validation loader drops its
    ragged last batch"""
    ds = ReferenceDataset(dir_src_img, dir_ref_img, dir_mask, identity_file, apply_transform=apply_transform, scale=img_scale, use_ssim=use_ssim,
                          device=device)
    n_val = math.ceil(len(ds) * val_amount)
    perm = torch.randperm(len(ds)).tolist()
    cut = len(ds) - n_val
    return (DeviceLoader(ds, perm[:cut], batch_size, shuffle=True, num_workers=num_workers),
            DeviceLoader(ds, perm[cut:], batch_size, shuffle=False, drop_last=True, num_workers=num_workers))


def to_device_batch(batch, device=None):
    """the boundary of the training loops (train_reference_fill.py:337-340, train_psp.py:309-312): every tensor on the device, plus
    ``true_masks = (mask > 0).float()`` from the bit-exact index kernel"""
    from . import functional as FF

    out = {k: (v if device is None or v.device == torch.device(device) else v.to(device, non_blocking=True)) for k, v in batch.items()}
    out["true_masks"] = FF.binarise_mask(out["mask"].contiguous())
    return out
