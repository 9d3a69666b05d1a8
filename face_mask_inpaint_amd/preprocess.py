"""Device-side image preprocessing for the data path (the reference's dataloader.py:76-93 ``preprocess`` and :169-170 ``Normalize``).

Split of work: the host decodes files and computes the O(W + H) resampling tables with exactly the double-precision arithmetic Pillow
uses (so the integer weights are Pillow's); every per-pixel operation -- both passes of the 8-bit BICUBIC resampling, the NEAREST
gather of the masks, uint8 HWC -> float32 CHW / 255 and the optional ``(x - 0.5) / 0.5`` -- is a kernel of libfmi_hip.so
(csrc/preproc.hip) working in integers / through a 256-entry table, so results equal the reference's CPU pipeline bit for bit
(tests/test_gpu_data.py against tests/golden/dataset.pt).  There is no CPU path: without the GPU ``DevicePreprocessor`` raises.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import FmiError

_FIXED_ONE = 1 << 22  # Pillow's PRECISION_BITS = 32 - 8 - 2


def _cubic(t: np.ndarray) -> np.ndarray:
    """Keys cubic convolution kernel with a = -0.5, evaluated in the operation order of Pillow's bicubic_filter"""
    a = -0.5
    t = np.abs(t)
    near = ((a + 2.0) * t - (a + 3.0)) * t * t + 1
    far = (((t - 5) * t + 8) * t - 4) * a
    return np.where(t < 1.0, near, np.where(t < 2.0, far, 0.0))


@lru_cache(maxsize=64)
def bicubic_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [out][2], weights int32 [out][ksize], ksize) of Pillow's 8-bit BICUBIC resampling from in_size to out_size samples.
    Vectorised over the output index; the window sum runs sequentially over the taps, as Pillow's does (a pairwise sum would round
    differently)."""
    ratio = float(in_size) / out_size
    stretch = ratio if ratio > 1.0 else 1.0
    reach = 2.0 * stretch
    ksize = int(np.ceil(reach)) * 2 + 1
    centre = (np.arange(out_size, dtype=np.float64) + 0.5) * ratio
    first = np.trunc(centre - reach + 0.5).astype(np.int64).clip(min=0)
    count = np.minimum(np.trunc(centre + reach + 0.5).astype(np.int64), in_size) - first
    tap = np.arange(ksize, dtype=np.int64)[None, :]
    live = tap < count[:, None]
    w = np.where(live, _cubic(((tap + first[:, None]).astype(np.float64) - centre[:, None] + 0.5) * (1.0 / stretch)), 0.0)
    total = np.zeros(out_size, np.float64)
    for j in range(ksize):
        total = total + w[:, j]
    w = np.where((total != 0.0)[:, None], w / np.where(total == 0.0, 1.0, total)[:, None], w)
    fixed = np.trunc(np.where(w < 0, -0.5, 0.5) + w * _FIXED_ONE).astype(np.int32)
    fixed[~live] = 0
    return np.stack([first, count], 1).astype(np.int32), np.ascontiguousarray(fixed), ksize


@lru_cache(maxsize=64)
def nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """source index of every output index for Pillow's NEAREST resize: the coordinate starts at step / 2 and is ADVANCED by repeated
    addition in double precision (np.add.accumulate adds sequentially), then truncated"""
    step = float(in_size) / out_size
    pos = np.add.accumulate(np.concatenate([[step * 0.5], np.full(out_size - 1, step)]))
    return np.trunc(pos).astype(np.int64).clip(0, in_size - 1).astype(np.int32)


@lru_cache(maxsize=4)
def _lut(normalise: bool) -> np.ndarray:
    t = (np.arange(256, dtype=np.float64) / 255).astype(np.float32)  # numpy: float64 quotient, then the cast (dataloader.py:89-90)
    if normalise:
        t = (t - np.float32(0.5)) / np.float32(0.5)  # torchvision Normalize([0.5] * 3, [0.5] * 3) in float32
    return t


def scaled_size(width: int, height: int, scale: float) -> Tuple[int, int]:
    nw, nh = int(scale * width), int(scale * height)
    if nw <= 0 or nh <= 0:
        raise ValueError("Scale is too small, resized images would have no pixel")
    return nw, nh


class DevicePreprocessor:
    """uint8 images / masks (host arrays of one common size) -> the tensors of the hot path, on ``device``."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise FmiError("DevicePreprocessor runs on the GPU; there is no CPU preprocessing path in this library")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise FmiError("DevicePreprocessor needs a cuda device")
        self._tables = {}

    def _dev(self, key, make):
        t = self._tables.get(key)
        if t is None:
            t = self._tables[key] = torch.from_numpy(np.ascontiguousarray(make())).to(self.device)
        return t

    @staticmethod
    def _stack(arrays: Sequence[np.ndarray]) -> np.ndarray:
        a0 = arrays[0]
        for a in arrays:
            if a.shape != a0.shape or a.dtype != np.uint8:
                raise FmiError("a batch is preprocessed in one launch: uint8 arrays of one common size")
        return np.ascontiguousarray(np.stack(arrays))

    def images(self, arrays: Sequence[np.ndarray], scale: float = 1.0, normalise: bool = False, also_plain: bool = False):
        """decoded uint8 [H][W][3] (or [H][W]) arrays -> float32 [N][C][H'][W'], BICUBIC-resized by ``scale``, / 255 (, Normalize).
        also_plain: returns (normalised, plain [0, 1]) from the one resized image (the gt_img / raw_gt_img pair of dataloader.py:249-254)"""
        host = self._stack([a if a.ndim == 3 else a[..., None] for a in arrays])
        n, h, w, c = host.shape
        nw, nh = scaled_size(w, h, scale)
        lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            cur = torch.from_numpy(host).to(self.device)
            cur_h = h
            vb = vk = None
            y_first, y_last = 0, h
            if nh != h:
                b, k, ks_v = bicubic_tables(h, nh)
                y_first, y_last = int(b[0, 0]), int(b[-1, 0] + b[-1, 1])
                shifted = b.copy()
                if nw != w:
                    shifted[:, 0] -= y_first  # the horizontal pass only produces the rows the vertical pass reads
                vb = self._dev(("vb", h, nh, nw != w), lambda: shifted)
                vk = self._dev(("vk", h, nh), lambda: k)
            if nw != w:
                b, k, ks_h = bicubic_tables(w, nw)
                rows = y_last - y_first
                out = torch.empty((n, rows, nw, c), device=self.device, dtype=torch.uint8)
                lib.resample_u8(cur.data_ptr(), out.data_ptr(), n, h, w, c, nw, 0, y_first, rows, self._dev(("hb", w, nw), lambda: b).data_ptr(),
                                self._dev(("hk", w, nw), lambda: k).data_ptr(), ks_h, st)
                cur, cur_h = out, rows
            if nh != h:
                out = torch.empty((n, nh, cur.shape[2], c), device=self.device, dtype=torch.uint8)
                lib.resample_u8(cur.data_ptr(), out.data_ptr(), n, cur_h, cur.shape[2], c, nh, 1, 0, 0, vb.data_ptr(), vk.data_ptr(), ks_v, st)
                cur = out
            res = torch.empty((n, c, nh, nw), device=self.device, dtype=torch.float32)
            lib.u8_lut_chw_f32(cur.data_ptr(), self._dev(("lut", normalise), lambda: _lut(normalise)).data_ptr(), res.data_ptr(), n, nh, nw, c, st)
            if also_plain and normalise:
                plain = torch.empty_like(res)
                lib.u8_lut_chw_f32(cur.data_ptr(), self._dev(("lut", False), lambda: _lut(False)).data_ptr(), plain.data_ptr(), n, nh, nw, c, st)
                return res, plain
        return (res, res) if also_plain else res

    def masks(self, arrays: Sequence[np.ndarray], scale: float = 1.0) -> torch.Tensor:
        """uint8 [H][W] binary maps -> int64 [N][H'][W'], NEAREST-resized by ``scale``"""
        host = self._stack(list(arrays))
        if host.ndim != 3:
            raise FmiError("binary maps are single-channel uint8 arrays")
        n, h, w = host.shape
        nw, nh = scaled_size(w, h, scale)
        lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            src = torch.from_numpy(host).to(self.device)
            out = torch.empty((n, nh, nw), device=self.device, dtype=torch.int64)
            lib.gather_u8_i64(src.data_ptr(), out.data_ptr(), n, h, w, nh, nw, self._dev(("ny", h, nh), lambda: nearest_table(h, nh)).data_ptr(),
                              self._dev(("nx", w, nw), lambda: nearest_table(w, nw)).data_ptr(), st)
        return out
