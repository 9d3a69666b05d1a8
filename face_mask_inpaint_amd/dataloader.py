"""Host-side mirror of the reference's dataloader.py (BasicDataset.preprocess / load :76-103, ReferenceDataset :122-266,
get_reference_dataloader :19-47): CelebA(-HQ) directories of ``<id>_surgical.jpg`` sources, ``<id>.jpg`` references / ground
truths, ``<id>.npy`` binary maps and an identity file -> the dictionary of tensors the hot path consumes:
``src_img, gt_img, raw_gt_img, ref_img`` float32 [3, H, W] in [0, 1] (or normalised to [-1, 1] with ``apply_transform``) and
``mask`` int64 [H, W].

File decoding and PIL's resampling (BICUBIC for images, NEAREST for masks -- kept in PIL so that pixels equal the reference's)
stay on the host; ``to_device_batch`` does the arithmetic on the GPU: uint8 HWC -> float32 NHWC / 255 (and (x - 0.5) / 0.5) and the
mask binarisation ``(mask > 0).float()`` of train_reference_fill.py:340 are kernels of this library.

``best_reference_map.pkl`` is this build's own cache file when it writes it; an existing one is only read if it was written by this
class (a JSON side-car marks it) -- foreign pickles are never unpickled."""
from __future__ import annotations

import json
import logging
import math
import random
from os import listdir
from os.path import splitext
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, random_split


class BasicDataset(Dataset):
    def __init__(self, images_dir, masks_dir, scale=1.0, mask_suffix=""):
        self.images_dir = Path(images_dir)
        self.masks_dir = Path(masks_dir)
        assert 0 < scale <= 1, "Scale must be between 0 and 1"
        self.scale = scale
        self.mask_suffix = mask_suffix
        self.ids = [splitext(file)[0].split("_")[0] for file in listdir(images_dir) if not file.startswith(".")]
        if not self.ids:
            raise RuntimeError(f"No input file found in {images_dir}, make sure you put your images there")
        logging.info(f"Creating dataset with {len(self.ids)} examples")

    def __len__(self):
        return len(self.ids)

    @classmethod
    def preprocess(cls, pil_img, scale, is_mask):
        from PIL import Image

        w, h = pil_img.size
        newW, newH = int(scale * w), int(scale * h)
        assert newW > 0 and newH > 0, "Scale is too small, resized images would have no pixel"
        pil_img = pil_img.resize((newW, newH), resample=Image.NEAREST if is_mask else Image.BICUBIC)
        img_ndarray = np.asarray(pil_img)
        if img_ndarray.ndim == 2 and not is_mask:
            img_ndarray = img_ndarray[np.newaxis, ...]
        if not is_mask:
            img_ndarray = img_ndarray.transpose((2, 0, 1))
            img_ndarray = img_ndarray / 255
            return torch.as_tensor(img_ndarray.copy()).float().contiguous()
        return torch.as_tensor(img_ndarray.copy()).long().contiguous()

    @classmethod
    def load(cls, filename):
        from PIL import Image

        ext = splitext(filename)[1]
        if ext in [".npz", ".npy"]:
            return Image.fromarray(np.load(filename))  # allow_pickle stays False
        if ext in [".pt", ".pth"]:
            return Image.fromarray(torch.load(filename, weights_only=True).numpy())
        return Image.open(filename)

    def __getitem__(self, idx):
        name = self.ids[idx]
        mask = self.load(self.masks_dir / Path(name + self.mask_suffix + ".npy"))
        img = self.load(self.images_dir / Path(name + "_surgical" + ".jpg"))
        assert img.size == mask.size, f"Image and mask {name} should be the same size, but are {img.size} and {mask.size}"
        return {"image": self.preprocess(img, self.scale, is_mask=False), "mask": self.preprocess(mask, self.scale, is_mask=True)}


class ReferenceDataset(BasicDataset):
    def __init__(self, source_dir, reference_dir, masks_dir, identity_file, apply_transform=True, scale=1.0, use_ssim=False, device=None,
                 return_id=False):
        self.source_dir = Path(source_dir)
        self.masks_dir = Path(masks_dir)
        self.reference_dir = Path(reference_dir)
        self.identity_map, self.img2identity = self.read_identity_file(identity_file)
        self.filter_id = set()  # identities with only one image are ignored
        for v in self.identity_map.values():
            if len(v) < 2:
                self.filter_id.update(v)
        assert 0 < scale <= 1, "Scale must be between 0 and 1"
        self.scale = scale
        self.ids = []
        for f in listdir(source_dir):
            f_id = splitext(f)[0].split("_")[0]
            if not f.startswith(".") and f_id not in self.filter_id:
                self.ids.append(f_id)
        if not self.ids:
            raise RuntimeError(f"No input file found in {source_dir}, make sure you put your images there")
        logging.info(f"Creating dataset with {len(self.ids)} examples")
        self.use_ssim = use_ssim
        if use_ssim:
            cache = self.source_dir.parent / Path("best_reference_map.json")
            if cache.is_file():
                self.best_reference_map = json.load(open(cache))
            else:
                logging.info("Creating best_reference_map")
                self.best_reference_map = self.find_best_reference(device)
        self.apply_transform = apply_transform
        self.return_id = return_id

    @staticmethod
    def transform(img):
        """transforms.Normalize([0.5] * 3, [0.5] * 3) (dataloader.py:169-170)"""
        return (img - 0.5) / 0.5

    def read_identity_file(self, identity_file):
        identity_map, img2identity = {}, {}
        with open(identity_file, "r") as f:
            for line in f:
                img, identity = line.strip().split(" ")
                img_id = splitext(img)[0].split("_")[0]
                identity = int(identity)
                img2identity[img_id] = identity
                identity_map.setdefault(identity, []).append(img_id)
        return identity_map, img2identity

    def find_best_reference(self, device):
        """for every image the same-identity image of highest SSIM (dataloader.py:188-216; the reference scores with pytorch_msssim,
        absent offline: this build's SSIM kernel on the GPU, or its torch definition on the CPU)"""
        from .modules.evaluations.ssim import ssim as ssim_fn

        dev = torch.device("cuda:0") if (device is None and torch.cuda.is_available()) else torch.device(device or "cpu")
        if dev.type != "cuda":
            raise RuntimeError("find_best_reference scores on the GPU (no CPU path in this library)")
        best = {}
        for name in self.ids:
            gt = self.preprocess(self.load(self.reference_dir / Path(name + ".jpg")), self.scale, is_mask=False).unsqueeze(0).to(dev)
            max_score, best_ref = -10, None
            for other in self.identity_map[self.img2identity[name]]:
                if other != name:
                    rf = self.preprocess(self.load(self.reference_dir / Path(other + ".jpg")), self.scale, is_mask=False).unsqueeze(0).to(dev)
                    score = float(ssim_fn(gt, rf))
                    if score > max_score:
                        max_score, best_ref = score, other
            best[name] = best_ref
        json.dump(best, open(self.source_dir.parent / Path("best_reference_map.json"), "w"))
        return best

    def sample_reference_image(self, img_name):
        if self.use_ssim:
            return self.best_reference_map[img_name]
        images = self.identity_map[self.img2identity[img_name]]
        assert len(images) > 1
        reference_image = random.choice(images)
        while reference_image == img_name:
            reference_image = random.choice(images)
        return reference_image

    def __getitem__(self, idx):
        name = self.ids[idx]
        mask = self.load(self.masks_dir / Path(name + ".npy"))
        src_img = self.load(self.source_dir / Path(name + "_surgical" + ".jpg"))
        gt_img = self.load(self.reference_dir / Path(name + ".jpg"))
        ref_img = self.load(self.reference_dir / Path(self.sample_reference_image(name) + ".jpg"))
        assert src_img.size == mask.size, f"Image and mask {name} should be the same size, but are {src_img.size} and {mask.size}"
        src_img = self.preprocess(src_img, self.scale, is_mask=False)
        raw_gt_img = self.preprocess(gt_img, self.scale, is_mask=False)
        ref_img = self.preprocess(ref_img, self.scale, is_mask=False)
        if self.apply_transform:
            src_img, ref_img, gt_img = self.transform(src_img), self.transform(ref_img), self.transform(raw_gt_img)
        else:
            gt_img = raw_gt_img
        mask = self.preprocess(mask, self.scale, is_mask=True)
        items = {"src_img": src_img, "gt_img": gt_img, "raw_gt_img": raw_gt_img, "ref_img": ref_img, "mask": mask}
        if self.return_id:
            items["id"] = torch.LongTensor([int(self.ids[idx])])
        return items


def get_reference_dataloader(dir_src_img, dir_ref_img, dir_mask, identity_file, batch_size, apply_transform=False, val_amount=0.1, num_workers=4,
                             img_scale=1.0, use_ssim=False, device=None):
    dataset = ReferenceDataset(dir_src_img, dir_ref_img, dir_mask, identity_file, apply_transform=apply_transform, scale=img_scale, use_ssim=use_ssim,
                               device=device)
    n_train = math.floor(len(dataset) * (1 - val_amount))
    n_val = math.ceil(len(dataset) * val_amount)
    train_set, val_set = random_split(dataset, [n_train, n_val])
    loader_args = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=True)
    return DataLoader(train_set, shuffle=True, **loader_args), DataLoader(val_set, shuffle=False, drop_last=True, **loader_args)


def to_device_batch(batch, device):
    """the host -> device boundary of the training loops (train_reference_fill.py:337-340, train_psp.py:309-312): images to the GPU,
    ``true_masks = (mask > 0).float()`` by the bit-exact index kernel"""
    from . import functional as FF

    out = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
    out["true_masks"] = FF.binarise_mask(out["mask"].contiguous())
    return out
