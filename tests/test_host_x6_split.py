"""The arithmetic behind csrc/x6.h, restated in numpy (no GPU): an fp32 value is the exact sum of three round-to-nearest bf16 pieces, and the six
piece products the kernels keep reproduce x * y to better than one fp32 rounding."""
import numpy as np


def _rn_bf16(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to bf16, returned as float32 (what v_cvt_pk_bf16_f32 computes for finite inputs)"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return r.astype(np.uint32).view(np.float32)


def _split3(x):
    x0 = _rn_bf16(x)
    r1 = (x - x0).astype(np.float32)  # exact: x0 keeps the leading 8 bits
    x1 = _rn_bf16(r1)
    x2 = (r1 - x1).astype(np.float32)
    return x0, x1, x2


def _values(n, seed):
    g = np.random.default_rng(seed)
    v = g.standard_normal(n).astype(np.float32) * np.exp(g.uniform(-60, 60, n)).astype(np.float32)
    edge = np.array([1.0, -1.0, 0.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.99998, 3.0e38, -2.0e-30, 2.0 ** -100, 1.5, 0.1, 16777215.0], np.float32)
    return np.concatenate([v, edge])


def test_three_bf16_pieces_are_exact():
    x = _values(200000, 0)
    x0, x1, x2 = _split3(x)
    assert np.array_equal(_rn_bf16(x2), x2), "the third piece is a bf16 number"
    s = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    assert np.array_equal(s.astype(np.float32), x) and np.array_equal(s, x.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(x1[nz]) <= 2.0 ** -8 * np.abs(x[nz])) and np.all(np.abs(x2[nz]) <= 2.0 ** -16 * np.abs(x[nz]))


def test_six_products_are_an_fp32_accurate_product():
    x, y = _values(100000, 1), _values(100000, 2)[::-1].copy()
    keep = (np.abs(x) > 1e-9) & (np.abs(y) > 1e-9) & (np.abs(x) < 1e9) & (np.abs(y) < 1e9)  # products and their pieces stay fp32-normal
    x, y = x[keep], y[keep]
    xs, ys = [p.astype(np.float64) for p in _split3(x)], [p.astype(np.float64) for p in _split3(y)]
    six = xs[0] * ys[0] + xs[0] * ys[1] + xs[1] * ys[0] + xs[0] * ys[2] + xs[1] * ys[1] + xs[2] * ys[0]
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -24, rel.max()  # the dropped x1 y2 + x2 y1 + x2 y2; an fp32 multiply alone rounds by up to 2^-24
    for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)):  # a product of two 8-bit significands is exact in fp32: the MFMA accumulator adds exact terms
        p = xs[i] * ys[j]
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)
