"""CPU: index algebra of the implicit-GEMM convolution family (taps, sub-pixel phases of the adjoint,
reflect padding, packed-weight rows, output mapping, fast division) checked without a GPU.

csrc/conv.hip + csrc/gemm.hip are compiled with g++ -DFMI_HOST_EMU: the operand loaders and epilogues are the
very code the GPU kernel runs, only the MFMA tile loop is replaced by a triple loop (csrc/gemm_core.h)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from face_mask_inpaint_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "face_mask_inpaint_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    out = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libfmi_emu.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-DFMI_HOST_EMU", "-x", "c++",
                           os.path.join(CSRC, "conv.hip"), os.path.join(CSRC, "gemm.hip"), "-o", so])
    return _lib.Library(so, strict=False)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def pack(w):
    k, c, kh, kw = w.shape
    wf = w.permute(2, 3, 1, 0).reshape(kh * kw, c, k).contiguous()
    wt = w.permute(2, 3, 0, 1).reshape(kh * kw, k, c).contiguous()
    return wf, wt


def desc(x, w, stride, pad, pad_mode=0, x_cs=None, y_cs=None, dil=1):
    n, c, h, wd = x.shape
    k, _, kh, kw = w.shape
    oh = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ow = (wd + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    return _lib.ConvDesc(n, h, wd, c, oh, ow, k, x_cs or c, y_cs or k, kh, kw, stride, pad, pad_mode, dil), oh, ow


CASES = [  # n, c, k, h, w, ksz, stride, pad, dilation
    (2, 8, 12, 9, 7, 3, 1, 1, 1), (1, 3, 8, 10, 11, 3, 1, 1, 1), (2, 16, 5, 6, 6, 1, 1, 0, 1), (2, 8, 4, 12, 10, 3, 2, 1, 1),
    (1, 4, 6, 9, 8, 4, 2, 1, 1), (2, 8, 1, 7, 7, 3, 1, 0, 1), (1, 6, 3, 8, 8, 3, 1, 1, 1), (1, 8, 8, 11, 9, 1, 2, 0, 1),
    # dilated 3x3 convolutions of modules/drn.py (stride 1, padding = dilation), and a 7x7 stem
    (2, 8, 12, 11, 9, 3, 1, 2, 2), (1, 16, 8, 13, 12, 3, 1, 4, 4), (1, 4, 6, 9, 9, 3, 1, 2, 2), (1, 3, 8, 12, 10, 7, 1, 3, 1),
]


@pytest.mark.parametrize("n,c,k,h,w,ksz,stride,pad,dil", CASES)
def test_conv_fwd_dgrad_wgrad(emu, n, c, k, h, w, ksz, stride, pad, dil):
    g = torch.Generator().manual_seed(h * 100 + c)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = torch.randn(k, c, ksz, ksz, generator=g, requires_grad=True)
    b = torch.randn(k, generator=g)
    y = F.conv2d(x, wt_, b, stride=stride, padding=pad, dilation=dil)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, oh, ow = desc(x, wt_, stride, pad, dil=dil)
    wf, wtp = pack(wt_.detach())
    xh, gyh = nhwc(x.detach()), nhwc(gy)
    res = torch.randn(n, oh, ow, k, generator=g)
    out = torch.full((n, oh, ow, k), float("nan"))
    emu.conv2d_fwd_f32(C.byref(d), ptr(xh), ptr(wf), ptr(b), ptr(res), ptr(out), 0, 1, 0, None)
    torch.testing.assert_close(out, nhwc(y.detach()) + res, rtol=1e-5, atol=1e-5)
    dx = torch.full((n, h, w, c), float("nan"))
    emu.conv2d_dgrad_f32(C.byref(d), ptr(gyh), ptr(wtp), None, None, ptr(dx), 1, 0, None)
    torch.testing.assert_close(dx, nhwc(x.grad), rtol=1e-5, atol=1e-5)
    dwf = torch.zeros_like(wf)
    emu.conv2d_wgrad_f32(C.byref(d), ptr(xh), ptr(gyh), ptr(dwf), None, 1, 0, None)
    torch.testing.assert_close(dwf, pack(wt_.grad)[0], rtol=1e-4, atol=1e-4)
    if (ksz * ksz * c) % 4 == 0:  # bias gradient as an extra row of ones in the same product
        dwf2, db = torch.zeros_like(wf), torch.zeros(k)
        emu.conv2d_wgrad_f32(C.byref(d), ptr(xh), ptr(gyh), ptr(dwf2), ptr(db), 1, 0, None)
        torch.testing.assert_close(dwf2, dwf, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(db, gy.sum((0, 2, 3)), rtol=1e-4, atol=1e-4)


def test_conv_transpose_is_the_adjoint(emu):
    """ConvTranspose2d(k3,s2,p1,op1) of base_function.py:326-341 = fmi_conv2d_dgrad on the conv view."""
    g = torch.Generator().manual_seed(0)
    cs, cb, hs, ws = 8, 4, 5, 7
    x = torch.randn(2, cs, hs, ws, generator=g, requires_grad=True)
    w = torch.randn(cs, cb, 3, 3, generator=g, requires_grad=True)  # torch ConvTranspose2d layout [in, out, kh, kw]
    b = torch.randn(cb, generator=g)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    big = torch.zeros(2, cb, 2 * hs, 2 * ws)
    d, oh, ow = desc(big, w, 2, 1)  # conv view: x = big image (cb ch), y = small image (cs ch), weight [K=cs][C=cb]
    assert (oh, ow) == (hs, ws)
    wf, wtp = pack(w.detach())
    out = torch.full((2, 2 * hs, 2 * ws, cb), float("nan"))
    xh, gyh = nhwc(x.detach()), nhwc(gy)
    emu.conv2d_dgrad_f32(C.byref(d), ptr(xh), ptr(wtp), ptr(b), None, ptr(out), 1, 0, None)
    torch.testing.assert_close(out, nhwc(y.detach()), rtol=1e-5, atol=1e-5)
    gx = torch.full((2, hs, ws, cs), float("nan"))
    emu.conv2d_fwd_f32(C.byref(d), ptr(gyh), ptr(wf), None, None, ptr(gx), 0, 1, 0, None)
    torch.testing.assert_close(gx, nhwc(x.grad), rtol=1e-5, atol=1e-5)
    dwf = torch.zeros_like(wf)
    emu.conv2d_wgrad_f32(C.byref(d), ptr(gyh), ptr(xh), ptr(dwf), None, 1, 0, None)
    torch.testing.assert_close(dwf, pack(w.grad)[0], rtol=1e-4, atol=1e-4)


def test_reflect_pad_and_channel_slices(emu):
    g = torch.Generator().manual_seed(1)
    xfull = torch.randn(2, 6, 7, 12, generator=g)  # NHWC with 12 channels, conv reads channels 4..11
    xs = xfull[..., 4:].permute(0, 3, 1, 2)
    w = torch.randn(3, 8, 3, 3, generator=g)
    ref = torch.tanh(F.conv2d(F.pad(xs, (1, 1, 1, 1), mode="reflect"), w))
    d, oh, ow = desc(xs, w, 1, 1, pad_mode=1, x_cs=12, y_cs=5)
    yfull = torch.zeros(2, oh, ow, 5)
    wf, _ = pack(w)
    xv = xfull.reshape(-1)[4:]
    yv = yfull.reshape(-1)[1:]
    emu.conv2d_fwd_f32(C.byref(d), C.c_void_p(xv.data_ptr()), ptr(wf), None, None, C.c_void_p(yv.data_ptr()), 1, 1, 0, None)
    torch.testing.assert_close(yfull[..., 1:4], nhwc(ref), rtol=1e-5, atol=1e-5)
    assert torch.all(yfull[..., 0] == 0) and torch.all(yfull[..., 4] == 0)


def test_per_sample_weights(emu):
    """ModulatedConv2d's grouped conv (stylegan2/model.py:271-277): sample n uses weight n."""
    g = torch.Generator().manual_seed(2)
    n, c, k, h = 3, 8, 4, 6
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(n, k, c, 3, 3, generator=g)
    ref = torch.cat([F.conv2d(x[i:i + 1], w[i], padding=1) for i in range(n)])
    d, oh, ow = desc(x, w[0], 1, 1)
    wf = torch.stack([pack(w[i])[0] for i in range(n)])
    out = torch.zeros(n, oh, ow, k)
    xh = nhwc(x)
    emu.conv2d_fwd_f32(C.byref(d), ptr(xh), ptr(wf), None, None, ptr(out), 0, n, wf[0].numel(), None)
    torch.testing.assert_close(out, nhwc(ref), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_dense_gemm_layouts(emu, ta, tb):
    g = torch.Generator().manual_seed(3)
    bsz, m, n, k = 2, 13, 10, 21
    a = torch.randn(bsz, m, k, generator=g)
    b = torch.randn(bsz, k, n, generator=g)
    c0 = torch.randn(bsz, m, n, generator=g)
    bias = torch.randn(n, generator=g)
    am = a.transpose(1, 2).contiguous() if ta else a.contiguous()
    bm = b.transpose(1, 2).contiguous() if tb else b.contiguous()
    sa = (1, m) if ta else (k, 1)
    sb = (1, k) if tb else (n, 1)
    c = c0.clone()
    emu.gemm_f32(ptr(am), ptr(bm), ptr(c), m, n, k, sa[0], sa[1], sb[0], sb[1], n, 1, bsz, m * k, k * n, m * n, 0.5, 2.0, ptr(bias), None)
    torch.testing.assert_close(c, 0.5 * (a @ b) + bias + 2.0 * c0, rtol=1e-5, atol=1e-5)


def test_fastdiv_exhaustive_small():
    # the same formula as csrc/gemm_core.h make_fastdiv/fdiv, checked over the ranges the kernels use
    def mk(d):
        l = 0
        while (1 << l) < d:
            l += 1
        m = ((1 << 32) * ((1 << l) - d)) // d + 1
        return m & 0xFFFFFFFF, min(l, 1), max(l - 1, 0)

    rng = np.random.default_rng(0)
    for d in list(range(1, 70)) + [224, 225, 1024, 1026, 50176, 1048576, 3 * 224 * 224, 2 ** 23 + 1]:
        m, s1, s2 = mk(d)
        ns = np.concatenate([np.arange(0, 5000), rng.integers(0, 2 ** 31, 20000), np.array([d - 1, d, 2 * d - 1, 2 ** 31 - 1])]).astype(np.uint64)
        t = (ns * np.uint64(m)) >> np.uint64(32)
        q = ((t + ((ns - t) >> np.uint64(s1))) & np.uint64(0xFFFFFFFF)) >> np.uint64(s2)
        assert np.array_equal(q, ns // np.uint64(d)), d
