"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the image resampling the reference's data path gets from Pillow.

The reference calls ``pil_img.resize((newW, newH), resample=Image.NEAREST if is_mask else Image.BICUBIC)`` (dataloader.py:76-82) and
divides by 255 (:89).  Pillow is a third-party dependency that is absent from /root/reference (un-pinned there: scripts/env_setup.sh
installs "pillow" without a version); the version importable in this image is 12.2.0.  What follows restates its published algorithm
for 8-bit images (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
Vertical_8bpc, bicubic_filter with a = -0.5; src/libImaging/Geometry.c: ImagingScaleAffine for NEAREST) with plain loops / numpy.
Pinned by (a) Pillow itself on random images of several sizes and scales (tests/test_oracle_data.py, runs where Pillow is
importable) and (b) tests/golden/dataset.pt, the tensors the reference's own ReferenceDataset returned for the committed jpg / npy
directory.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: 8 bits of data, 2 bits of head room for the overshoot of the cubic


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int, support0: float = 2.0):
    """bounds[xx] = (first input index, count), kk[xx][0..ksize) integer weights (sum ~ 2^22) of output index xx"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img uint8 [H][W][C] -> [out_h][out_w][C]: horizontal pass into an 8-bit intermediate (only the rows the vertical pass reads), then
    the vertical pass -- each output = clip8((2^21 + sum in * k) >> 22)"""
    h, w = img.shape[:2]
    src = img.astype(np.int64)
    need_h, need_v = out_w != w, out_h != h
    if need_v:
        bv, kv, _ = precompute_coeffs(h, out_h)
        y_first, y_last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
    else:
        y_first, y_last = 0, h
    if need_h:
        bh, kh, _ = precompute_coeffs(w, out_w)
        tmp = np.zeros((y_last - y_first, out_w) + img.shape[2:], np.uint8)
        for xx in range(out_w):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = np.full((y_last - y_first,) + img.shape[2:], 1 << (PRECISION_BITS - 1), np.int64)
            for x in range(n):
                acc += src[y_first:y_last, x0 + x] * int(kh[xx, x])
            tmp[:, xx] = _clip8(acc)
        src = tmp.astype(np.int64)
        row_off = y_first
    else:
        row_off = 0
    if not need_v:
        return src.astype(np.uint8)
    out = np.zeros((out_h, src.shape[1]) + img.shape[2:], np.uint8)
    for yy in range(out_h):
        y0, n = int(bv[yy, 0]) - row_off, int(bv[yy, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for y in range(n):
            acc += src[y0 + y] * int(kv[yy, y])
        out[yy] = _clip8(acc)
    return out


def nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """Geometry.c ImagingScaleAffine: source index of every output index, the coordinate ACCUMULATED in double as Pillow does"""
    a = float(in_size) / out_size
    o = a * 0.5
    tab = np.zeros(out_size, np.int64)
    for x in range(out_size):
        tab[x] = -1 if o < 0.0 else int(o)
        o += a
    return np.clip(tab, 0, in_size - 1)  # indices outside the image are never produced for a full-image box


def resize_nearest(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    h, w = img.shape[:2]
    return img[nearest_table(h, out_h)][:, nearest_table(w, out_w)]


def preprocess(img_u8: np.ndarray, scale: float, is_mask: bool) -> np.ndarray:
    """dataloader.py:76-93 on a decoded array: images -> float32 [C][H'][W'] in [0, 1] (float64 division, then the cast), masks -> int64"""
    h, w = img_u8.shape[:2]
    nw, nh = int(scale * w), int(scale * h)
    assert nw > 0 and nh > 0
    if is_mask:
        return resize_nearest(img_u8, nw, nh).astype(np.int64)
    r = resize_bicubic_u8(img_u8 if img_u8.ndim == 3 else img_u8[..., None], nw, nh)
    return (r.transpose(2, 0, 1) / 255).astype(np.float32)
