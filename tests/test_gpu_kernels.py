"""GPU: every HIP kernel through the C ABI against plain torch fp32 (CPU) on the same seeded inputs.
Tolerances are stated per test; index / integer work is bit exact."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def FF():
    from face_mask_inpaint_amd import functional

    return functional


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def pack(w):
    k, c, kh, kw = w.shape
    return (w.permute(2, 3, 1, 0).reshape(kh * kw, c, k).contiguous(), w.permute(2, 3, 0, 1).reshape(kh * kw, k, c).contiguous())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k,bsz", [(13, 10, 21, 2), (128, 128, 64, 1), (257, 100, 130, 3), (64, 300, 48, 2), (30, 200, 17, 1), (500, 24, 260, 2)])
def test_gemm(dev, FF, ta, tb, m, n, k, bsz):
    """fp32 MFMA = fmaf chain: error ~1e-7 * sum|a b| -> rtol 1e-5 at K <= 260"""
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(bsz, m, k, generator=g)
    b = torch.randn(bsz, k, n, generator=g)
    c0 = torch.randn(bsz, m, n, generator=g)
    bias = torch.randn(n, generator=g)
    am = (a.transpose(1, 2).contiguous() if ta else a.contiguous()).to(dev)
    bm = (b.transpose(1, 2).contiguous() if tb else b.contiguous()).to(dev)
    sa = (1, m) if ta else (k, 1)
    sb = (1, k) if tb else (n, 1)
    c = c0.clone().to(dev)
    biasd = bias.to(dev)
    FF.gemm_raw(FF._p(am), FF._p(bm), FF._p(c), m, n, k, sa, sb, (n, 1), bsz, (m * k, k * n, m * n), 0.5, 2.0, biasd)
    ref = 0.5 * (a.double() @ b.double()) + bias.double() + 2.0 * c0.double()
    torch.testing.assert_close(c.cpu().double(), ref, rtol=1e-5, atol=1e-4)


def test_gemm_split_k_skinny(dev, FF):
    """skinny output + long reduction takes the split-K / fp32-atomic path (attention P.V and dS.K shapes)"""
    g = torch.Generator().manual_seed(77)
    for (m, n, k, bsz, beta) in [(128, 256, 4096, 2, 0.0), (100, 64, 8192, 1, 1.0), (128, 40, 2048, 3, 0.5)]:
        a = torch.randn(bsz, m, k, generator=g) / k ** 0.5
        b = torch.randn(bsz, k, n, generator=g)
        c0 = torch.randn(bsz, m, n, generator=g)
        bias = torch.randn(n, generator=g)
        ad, bd, c, biasd = a.to(dev), b.to(dev), c0.clone().to(dev), bias.to(dev)
        FF.gemm_raw(FF._p(ad), FF._p(bd), FF._p(c), m, n, k, (k, 1), (n, 1), (n, 1), bsz, (m * k, k * n, m * n), 2.0, beta, biasd)
        ref = 2.0 * (a.double() @ b.double()) + bias.double() + beta * c0.double()
        torch.testing.assert_close(c.cpu().double(), ref, rtol=1e-5, atol=1e-4)


def test_gemm_mfma_layout_asymmetric(dev, FF):
    """A = I against an asymmetric integer B catches swapped row/column fragment maps exactly."""
    n = 96
    a = torch.eye(n)
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 5.0
    c = torch.zeros(n, n, device=dev)
    ad, bd = a.to(dev), b.to(dev)
    FF.gemm_raw(FF._p(ad), FF._p(bd), FF._p(c), n, n, n, (n, 1), (n, 1), (n, 1))
    assert torch.equal(c.cpu(), b)


CONV_CASES = [  # n, c, k, h, w, ksz, stride, pad
    (2, 8, 12, 9, 7, 3, 1, 1), (1, 3, 8, 10, 11, 3, 1, 1), (2, 16, 5, 6, 6, 1, 1, 0), (2, 8, 4, 12, 10, 3, 2, 1),
    (1, 4, 6, 9, 8, 4, 2, 1), (2, 8, 1, 7, 7, 3, 1, 0), (2, 32, 64, 33, 31, 3, 1, 1), (1, 64, 32, 40, 40, 3, 1, 1),
    (2, 128, 128, 16, 16, 3, 1, 1), (1, 32, 3, 64, 64, 3, 1, 1), (2, 256, 64, 8, 8, 1, 1, 0),
    # LDS-DMA pipeline (C % 16 == 0, K % 4 == 0): one-tile reductions (1x1, C = 16), two-tile, narrow 128x32 / 32x128 /
    # 64-wide tiles, stride-2 adjoints, tails in M and N, enough rows for several workgroups per CU
    (2, 16, 16, 32, 32, 1, 1, 0), (2, 16, 16, 16, 16, 1, 1, 0), (2, 32, 16, 32, 32, 1, 1, 0), (2, 16, 64, 32, 32, 1, 1, 0),
    (2, 16, 128, 20, 20, 1, 1, 0), (2, 16, 48, 33, 31, 3, 1, 1), (2, 32, 32, 16, 16, 3, 1, 1), (3, 64, 36, 20, 20, 3, 2, 1),
    (2, 16, 32, 32, 32, 3, 2, 1), (2, 32, 64, 32, 32, 3, 2, 1), (2, 16, 16, 32, 32, 4, 2, 1), (2, 48, 64, 17, 19, 5, 1, 2),
    (4, 32, 32, 128, 128, 3, 1, 1), (2, 256, 256, 24, 24, 3, 1, 1), (1, 512, 128, 14, 14, 3, 1, 1),
    # stride-2 adjoints of small maps with a deep reduction (the pSp style heads, psp_encoders.py:13-36): split over workgroups, all
    # sub-pixel phases add into one initialised output (incl. phases of odd extents and the 2x2 -> 1x1 case)
    (4, 512, 256, 8, 8, 3, 2, 1), (2, 256, 512, 5, 7, 3, 2, 1), (8, 512, 512, 2, 2, 3, 2, 1),
    # dilated 3x3 convolutions of modules/drn.py (padding = dilation; a 9th entry = the dilation) on the LDS-DMA and the register path,
    # and the 7x7 / 11x11-stride-4 stems of DRN and LPIPS' AlexNet
    (2, 256, 256, 8, 6, 3, 1, 2, 2), (2, 512, 512, 8, 6, 3, 1, 4, 4), (2, 64, 48, 17, 15, 3, 1, 2, 2), (2, 8, 12, 11, 9, 3, 1, 2, 2),
    (2, 3, 16, 64, 48, 7, 1, 3), (2, 3, 64, 63, 63, 11, 4, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_family(dev, FF, case):
    from face_mask_inpaint_amd import _lib

    n, c, k, h, w, ksz, stride, pad = case[:8]
    dil = case[8] if len(case) > 8 else 1
    lib = _lib.lib()
    g = torch.Generator().manual_seed(h * 100 + c)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = (torch.randn(k, c, ksz, ksz, generator=g) / (c * ksz * ksz) ** 0.5).requires_grad_(True)
    b = torch.randn(k, generator=g)
    y = F.conv2d(x, wt_, b, stride=stride, padding=pad, dilation=dil)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, oh, ow = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, dil=dil)
    wf, wtp = [t.to(dev) for t in pack(wt_.detach())]
    xh, gyh, bd = nhwc(x.detach()).to(dev), nhwc(gy).to(dev), b.to(dev)
    res = torch.randn(n, oh, ow, k, generator=g)
    resd = res.to(dev)
    out = torch.full((n, oh, ow, k), float("nan"), device=dev)
    st = FF._st()
    lib.conv2d_fwd_f32(C.byref(d), FF._p(xh), FF._p(wf), FF._p(bd), FF._p(resd), FF._p(out), 0, 1, 0, st)
    torch.testing.assert_close(out.cpu(), nhwc(y.detach()) + res, rtol=1e-4, atol=1e-5)
    dx = torch.full((n, h, w, c), float("nan"), device=dev)
    lib.conv2d_dgrad_f32(C.byref(d), FF._p(gyh), FF._p(wtp), None, None, FF._p(dx), 1, 0, st)
    torch.testing.assert_close(dx.cpu(), nhwc(x.grad), rtol=1e-4, atol=1e-5)
    dwf = torch.zeros_like(wf)
    lib.conv2d_wgrad_f32(C.byref(d), FF._p(xh), FF._p(gyh), FF._p(dwf), None, 1, 0, st)
    watol = 2e-4 * max(1.0, (n * oh * ow / 2048.0) ** 0.5)  # fp32 sums over all pixels, on both sides: error grows like sqrt(#terms)
    torch.testing.assert_close(dwf.cpu(), pack(wt_.grad)[0], rtol=1e-4, atol=watol)
    db = torch.zeros(k, device=dev)
    lib.bias_grad_f32(FF._p(gyh), n * oh * ow, k, k, FF._p(db), st)
    torch.testing.assert_close(db.cpu(), gy.sum((0, 2, 3)), rtol=1e-4, atol=watol)
    if (ksz * ksz * c) % 4 == 0:  # fused: the bias gradient is an extra row of the weight-gradient GEMM
        dwf2, db2 = torch.zeros_like(wf), torch.zeros(k, device=dev)
        lib.conv2d_wgrad_f32(C.byref(d), FF._p(xh), FF._p(gyh), FF._p(dwf2), FF._p(db2), 1, 0, st)
        torch.testing.assert_close(dwf2.cpu(), pack(wt_.grad)[0], rtol=1e-4, atol=watol)
        torch.testing.assert_close(db2.cpu(), gy.sum((0, 2, 3)), rtol=1e-4, atol=watol)


def test_conv_transpose_and_autograd(dev, FF):
    g = torch.Generator().manual_seed(0)
    cs, cb, hs, ws = 16, 8, 9, 11
    x = torch.randn(2, cs, hs, ws, generator=g, requires_grad=True)
    w = (torch.randn(cs, cb, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(cb, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    wd = w.detach().to(dev).requires_grad_(True)
    bd = b.detach().to(dev).requires_grad_(True)
    (pw,) = FF.prepare_weights([(wd, None, None)])
    out = FF.conv_transpose2d(xd, pw, bd)
    torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-4, atol=1e-5)
    out.backward(nhwc(gy).to(dev))
    torch.testing.assert_close(xd.grad.cpu(), nhwc(x.grad), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=1e-4)


def test_reflect_conv_tanh_autograd(dev, FF):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 9, 7, generator=g, requires_grad=True)
    w = (torch.randn(3, 8, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(3, generator=g, requires_grad=True)
    y = torch.tanh(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    wd, bd = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    (pw,) = FF.prepare_weights([(wd, None, None)])
    out = FF.conv2d(xd, pw, bd, None, 1, 1, 1, FF.ACT_TANH)
    torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-4, atol=1e-5)
    out.backward(nhwc(gy).to(dev))
    torch.testing.assert_close(xd.grad.cpu(), nhwc(x.grad), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=1e-4)


def test_spectral_norm_prepare_and_grad(dev, FF):
    """one power iteration + W/sigma + packing, and the gradient through sigma (external_function.py:30-41)"""
    g = torch.Generator().manual_seed(2)
    for shape in [(16, 8, 3, 3), (8, 24, 1, 1), (64, 32, 3, 3)]:
        w = torch.randn(*shape, generator=g, requires_grad=True)
        h = shape[0]
        u0 = F.normalize(torch.randn(h, generator=g), dim=0)
        v0 = F.normalize(torch.randn(w[0].numel(), generator=g), dim=0)
        wm = w.detach().reshape(h, -1)
        v1 = wm.t().mv(u0)
        v1 = v1 / (v1.norm() + 1e-12)
        u1 = wm.mv(v1)
        u1 = u1 / (u1.norm() + 1e-12)
        sigma = u1.dot(w.reshape(h, -1).mv(v1))
        weff = w / sigma
        gw = torch.randn(*shape, generator=g)
        weff.backward(gw)
        wd = w.detach().to(dev).requires_grad_(True)
        ud, vd = u0.to(dev), v0.to(dev)
        (pw,) = FF.prepare_weights([(wd, ud, vd)])
        torch.testing.assert_close(ud.cpu(), u1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(vd.cpu(), v1, rtol=1e-5, atol=1e-6)
        wf_ref, wt_ref = pack(weff.detach())
        torch.testing.assert_close(pw.wf.detach().cpu(), wf_ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pw.wt.cpu(), wt_ref, rtol=1e-5, atol=1e-6)
        pw.wf.backward(pack(gw)[0].to(dev))
        torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-5)


def test_spectral_norm_power_iterations(dev, FF):
    """SpectralNorm(power_iterations = k) (external_function.py:22,36: the loop runs k times per forward; every caller in the reference uses 1):
    one weight-preparation call with entries of different k, against the loop written out"""
    from face_mask_inpaint_amd.modules.pluralistic_model.external_function import SpectralNorm

    g = torch.Generator().manual_seed(12)
    items, refs = [], []
    for shape, k in (((16, 8, 3, 3), 3), ((32, 16, 1, 1), 1), ((24, 32, 3, 3), 5)):
        w = torch.randn(*shape, generator=g)
        h = shape[0]
        u = F.normalize(torch.randn(h, generator=g), dim=0)
        v = F.normalize(torch.randn(w[0].numel(), generator=g), dim=0)
        wm = w.reshape(h, -1)
        ur, vr = u.clone(), v.clone()
        for _ in range(k):
            vr = wm.t().mv(ur)
            vr = vr / (vr.norm() + 1e-12)
            ur = wm.mv(vr)
            ur = ur / (ur.norm() + 1e-12)
        refs.append((ur, vr, w / ur.dot(wm.mv(vr))))
        items.append((w.to(dev), u.to(dev), v.to(dev), True, k))
    for pw, it, (ur, vr, weff) in zip(FF.prepare_weights(items), items, refs):
        torch.testing.assert_close(it[1].cpu(), ur, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(it[2].cpu(), vr, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pw.wf.detach().cpu(), pack(weff)[0], rtol=1e-5, atol=1e-6)
    sn = SpectralNorm(torch.nn.Conv2d(8, 16, 3, padding=1), power_iterations=2).to(dev)
    w0, u0, v0 = sn.module.weight_bar.detach().clone(), sn.module.weight_u.detach().clone(), sn.module.weight_v.detach().clone()
    x = torch.randn(2, 8, 9, 7, generator=g).to(dev)
    y = sn(x)
    wm = w0.reshape(16, -1)
    for _ in range(2):
        v0 = wm.t().mv(u0)
        v0 = v0 / (v0.norm() + 1e-12)
        u0 = wm.mv(v0)
        u0 = u0 / (u0.norm() + 1e-12)
    ref = F.conv2d(x, w0 / u0.dot(wm.mv(v0)), sn.module.bias, padding=1)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sn.module.weight_u.detach(), u0, rtol=1e-5, atol=1e-6)


def test_eltwise_and_softplus(dev, FF):
    g = torch.Generator().manual_seed(3)
    a = torch.randn(3, 5, 7, 9, generator=g) * 3
    b = torch.randn(3, 5, 7, 9, generator=g)
    ad, bd = a.to(dev), b.to(dev)
    cases = {
        FF.EW_LRELU: F.leaky_relu(a, 0.1), FF.EW_LRELU_BWD: a * torch.where(b > 0, 1.0, 0.1), FF.EW_TANH_BWD: a * (1 - b * b),
        FF.EW_ADD: a + b, FF.EW_SCALE: a * 0.1, FF.EW_AXPY: 0.1 * a + b, FF.EW_MUL: a * b, FF.EW_RELU_BWD_OUT: a * (b > 0),
        FF.EW_SOFTPLUS: F.softplus(a), FF.EW_SOFTPLUS_BWD: a * torch.sigmoid(b), FF.EW_SUB: a - b,
    }
    for op, ref in cases.items():
        out = FF.eltwise(op, ad, bd, 0.1)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6, msg=lambda m, op=op: f"op {op}: {m}")
    big = torch.tensor([25.0, -30.0, 19.99, 20.01], device=dev)
    torch.testing.assert_close(FF.eltwise(FF.EW_SOFTPLUS, big).cpu(), F.softplus(big.cpu()), rtol=1e-6, atol=1e-7)


def test_pools_and_resize(dev, FF):
    g = torch.Generator().manual_seed(4)
    for c in (8, 3):
        x = torch.randn(2, c, 12, 8, generator=g, requires_grad=True)
        for k in (2, 4):
            y = F.avg_pool2d(x, k, k)
            gy = torch.randn(y.shape, generator=g)
            (gx,) = torch.autograd.grad(y, x, gy)
            xd = nhwc(x.detach()).to(dev).requires_grad_(True)
            out = FF.avg_pool(xd, k)
            torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-6, atol=1e-6)
            out.backward(nhwc(gy).to(dev))
            torch.testing.assert_close(xd.grad.cpu(), nhwc(gx), rtol=1e-6, atol=1e-6)
        y = F.max_pool2d(x, 2, 2)
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, x, gy)
        xd = nhwc(x.detach()).to(dev).requires_grad_(True)
        out = FF.max_pool2(xd)
        assert torch.equal(out.detach().cpu(), nhwc(y.detach()))
        out.backward(nhwc(gy).to(dev))
        torch.testing.assert_close(xd.grad.cpu(), nhwc(gx), rtol=0, atol=0)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = torch.rand(2, 3, 40, 36, generator=g, requires_grad=True)
    for oh, ow in ((28, 28), (40, 36), (5, 7), (50, 44)):
        y = (F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True) - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, x, gy)
        xd = nhwc(x.detach()).to(dev).requires_grad_(True)
        out = FF.resize_bilinear(xd, oh, ow, mean.to(dev), std.to(dev))
        torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-5, atol=1e-5)
        out.backward(nhwc(gy).to(dev))
        torch.testing.assert_close(xd.grad.cpu(), nhwc(gx), rtol=1e-4, atol=1e-5)


def test_mask_ops_bit_exact(dev, FF, golden):
    fx = golden("picnet_ops.pt")["binarise"]
    assert torch.equal(FF.binarise_mask(fx["mask"].to(dev)).cpu(), fx["out"])
    g = torch.Generator().manual_seed(5)
    m = torch.randint(-(2 ** 40), 2 ** 40, (3, 17, 19), generator=g)
    m[0, 0, :5] = torch.tensor([0, 1, -1, 255, -(2 ** 62)])
    assert torch.equal(FF.binarise_mask(m.to(dev)).cpu(), (m > 0).float())
    sc = golden("picnet_ops.pt")["scale_img"]
    md = sc["mask"].to(dev).permute(0, 2, 3, 1).contiguous()
    torch.testing.assert_close(FF.resize_bilinear(md, 4, 4).cpu().permute(0, 3, 1, 2), sc["out"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(FF.resize_bilinear(md, 5, 7).cpu().permute(0, 3, 1, 2), sc["out_odd"], rtol=1e-6, atol=1e-6)
    x = torch.randn(2, 6, 5, 3, generator=g)
    mk = (torch.rand(2, 6, 5, generator=g) < 0.5).float()
    assert torch.equal(FF.mask_mul(x.to(dev), mk.to(dev), True).cpu(), x * (1 - mk).unsqueeze(-1))
    assert torch.equal(FF.mask_mul(x.to(dev), mk.to(dev), False).cpu(), x * mk.unsqueeze(-1))


def test_instance_norm_act(dev, FF):
    g = torch.Generator().manual_seed(6)
    for (n, c, h, w, slope) in [(2, 8, 7, 9, 0.1), (2, 64, 33, 31, 0.1), (1, 32, 64, 64, 1.0), (3, 256, 5, 5, 0.1)]:
        x = (torch.randn(n, c, h, w, generator=g) * 2 + 0.7).requires_grad_(True)
        ga = (torch.randn(c, generator=g) * 0.5 + 1).requires_grad_(True)
        be = (torch.randn(c, generator=g) * 0.5).requires_grad_(True)
        y = F.leaky_relu(F.instance_norm(x, weight=ga, bias=be, eps=1e-5), slope)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xd = nhwc(x.detach()).to(dev).requires_grad_(True)
        gd, bd = ga.detach().to(dev).requires_grad_(True), be.detach().to(dev).requires_grad_(True)
        out = FF.instance_norm_act(xd, gd, bd, 1e-5, slope)
        torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-4, atol=1e-5)
        out.backward(nhwc(gy).to(dev))
        torch.testing.assert_close(xd.grad.cpu(), nhwc(x.grad), rtol=1e-3, atol=2e-5)
        torch.testing.assert_close(gd.grad.cpu(), ga.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bd.grad.cpu(), be.grad, rtol=1e-4, atol=1e-4)


def test_softmax_rows(dev, FF):
    from face_mask_inpaint_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(7)
    for rows, cols in [(5, 64), (3, 1000), (2, 16384), (4, 37)]:
        x = torch.randn(rows, cols, generator=g) * 4
        x[0, 3] = 60.0  # forces a large running-max jump
        p = torch.softmax(x, -1)
        dp = torch.randn(rows, cols, generator=g)
        ds = p * (dp - (p * dp).sum(-1, keepdim=True))
        xd, dpd = x.to(dev), dp.to(dev)
        pd = torch.empty_like(xd)
        lib.softmax_rows_f32(FF._p(xd), FF._p(pd), rows, cols, FF._st())
        torch.testing.assert_close(pd.cpu(), p, rtol=1e-5, atol=1e-8)
        lib.softmax_rows_bwd_f32(FF._p(pd), FF._p(dpd), FF._p(dpd), rows, cols, FF._st())
        torch.testing.assert_close(dpd.cpu(), ds, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n,t,d,cs", [(2, 64, 8, (16, 16)), (1, 300, 16, (32,)), (2, 1024, 32, (128, 128)), (1, 256, 64, (256,)),
                                      (2, 128, 32, (128,)), (1, 384, 64, (64, 64))])
def test_self_attention(dev, FF, n, t, d, cs, monkeypatch):
    """softmax(q q^T) v, forward and backward, including the multi-chunk path"""
    monkeypatch.setattr(FF, "ATTN_CHUNK_BYTES", t * 4 * 128 * 2)  # force several query chunks and image groups
    g = torch.Generator().manual_seed(8)
    q = (torch.randn(n, t, d, generator=g) * 0.7).requires_grad_(True)
    vs = [torch.randn(n, t, c, generator=g).requires_grad_(True) for c in cs]
    att = torch.softmax(q @ q.transpose(1, 2), -1)
    outs = [att @ v for v in vs]
    gos = [torch.randn(o.shape, generator=g) for o in outs]
    torch.autograd.backward(outs, gos)
    qd = q.detach().to(dev).requires_grad_(True)
    vds = [v.detach().to(dev).requires_grad_(True) for v in vs]
    res = FF.self_attention(qd, vds)
    for r, o in zip(res, outs):
        torch.testing.assert_close(r.detach().cpu(), o.detach(), rtol=1e-4, atol=1e-5)
    torch.autograd.backward(res, [go.to(dev) for go in gos])
    torch.testing.assert_close(qd.grad.cpu(), q.grad, rtol=1e-3, atol=3e-4)  # |grad| ~ 1..10, 1024-term fp32 sums on both sides
    for vd, v in zip(vds, vs):
        torch.testing.assert_close(vd.grad.cpu(), v.grad, rtol=1e-4, atol=1e-5)


def test_small_fused_ops(dev, FF):
    g = torch.Generator().manual_seed(9)
    n, h, w, z = 2, 4, 5, 8
    o_s, o_r = [torch.randn(n, h, w, 2 * z, generator=g).requires_grad_(True) for _ in range(2)]
    eq, ep = torch.randn(n, h, w, z, generator=g), torch.randn(n, h, w, z, generator=g)
    zq = o_s[..., :z] + F.softplus(o_s[..., z:]) * eq
    zp = o_r[..., :z] + F.softplus(o_r[..., z:]) * ep
    zz = torch.cat([zq, zp], -1)
    gz = torch.randn(zz.shape, generator=g)
    zz.backward(gz)
    sd, rd = o_s.detach().to(dev).requires_grad_(True), o_r.detach().to(dev).requires_grad_(True)
    out = FF.vae_sample(sd, rd, eq.to(dev), ep.to(dev))
    torch.testing.assert_close(out.detach().cpu(), zz.detach(), rtol=1e-6, atol=1e-6)
    out.backward(gz.to(dev))
    torch.testing.assert_close(sd.grad.cpu(), o_s.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rd.grad.cpu(), o_r.grad, rtol=1e-5, atol=1e-6)
    # guide blend + cat
    c = 6
    ra, rf, sa = [torch.randn(n, h, w, c, generator=g).requires_grad_(True) for _ in range(3)]
    m = torch.rand(n, h, w, generator=g)
    ref = torch.cat([(1 - m).unsqueeze(-1) * ra + m.unsqueeze(-1) * rf, sa], -1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    rad, rfd, sad = [t.detach().to(dev).requires_grad_(True) for t in (ra, rf, sa)]
    out = FF.guide_blend_cat(rad, rfd, sad, m.to(dev))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    out.backward(go.to(dev))
    for a, b in ((rad, ra), (rfd, rf), (sad, sa)):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-6, atol=1e-6)
    # gamma * o + x
    o, x = [torch.randn(n, h, w, c, generator=g).requires_grad_(True) for _ in range(2)]
    gam = torch.tensor([0.37], requires_grad=True)
    y = gam * o + x
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    od, xd, gd = o.detach().to(dev).requires_grad_(True), x.detach().to(dev).requires_grad_(True), gam.detach().to(dev).requires_grad_(True)
    out = FF.scale_add_param(od, gd, xd)
    torch.testing.assert_close(out.detach().cpu(), y.detach(), rtol=1e-6, atol=1e-6)
    out.backward(gy.to(dev))
    torch.testing.assert_close(od.grad.cpu(), o.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gd.grad.cpu(), gam.grad, rtol=1e-5, atol=1e-5)


def test_losses_against_golden(dev, FF, golden):
    fx = golden("picnet_ops.pt")
    xg = fx["gram"]["x"]
    n, c, h, w = xg.shape
    xd = nhwc(xg).to(dev).view(n, h * w, c)
    torch.testing.assert_close(FF.gram_matrix(xd).cpu(), fx["gram"]["out"], rtol=1e-5, atol=1e-7)
    from face_mask_inpaint_amd.modules.pluralistic_model import external_function as ef

    for name, fn in (("style_loss", ef.StyleLoss), ("contextual_loss", ef.contextual_loss)):
        x = fx[name]["x"].to(dev).requires_grad_(True)
        l = fn(x, fx[name]["y"].to(dev))
        torch.testing.assert_close(l.detach().cpu(), fx[name]["out"], rtol=1e-5, atol=1e-8)
        l.backward()
        torch.testing.assert_close(x.grad.cpu(), fx[name]["gx"], rtol=2e-4, atol=1e-8)
    gl = ef.GANLoss("lsgan")
    p = fx["lsgan"]["pred"].to(dev)
    torch.testing.assert_close(gl(p, True, True).cpu(), fx["lsgan"]["real"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gl(p, False, True).cpu(), fx["lsgan"]["fake"], rtol=1e-6, atol=1e-7)
    g = torch.Generator().manual_seed(10)
    a = torch.randn(2, 3, 17, 19, generator=g, requires_grad=True)
    b = torch.randn(2, 3, 17, 19, generator=g)
    for mine, ref in ((FF.l1_loss, F.l1_loss), (FF.mse_loss, F.mse_loss)):
        l = ref(a, b)
        (ga,) = torch.autograd.grad(l, a)
        ad = a.detach().to(dev).requires_grad_(True)
        lm = mine(ad, b.to(dev))
        torch.testing.assert_close(lm.detach().cpu(), l.detach(), rtol=1e-6, atol=1e-8)
        (lm * 3.0).backward()
        torch.testing.assert_close(ad.grad.cpu(), 3.0 * ga, rtol=1e-6, atol=1e-9)


def test_contextual_loss_larger(dev, FF):
    """28x28 feature map as at full size (784 points), random features"""
    g = torch.Generator().manual_seed(11)
    from oracle import picnet_cpu as O  # checker

    x = torch.randn(2, 24, 12, 12, generator=g).abs().requires_grad_(True)
    y = torch.randn(2, 24, 12, 12, generator=g).abs()
    l = O.contextual_loss(x, y)
    l.backward()
    from face_mask_inpaint_amd.modules.pluralistic_model import external_function as ef

    xd = x.detach().to(dev).requires_grad_(True)
    lm = ef.contextual_loss(xd, y.to(dev))
    torch.testing.assert_close(lm.detach().cpu(), l.detach(), rtol=1e-5, atol=1e-7)
    lm.backward()
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-3, atol=1e-8)


def test_fused_adam_matches_torch(dev):
    from face_mask_inpaint_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(12)
    shapes = [(7,), (33, 5), (4, 3, 3, 3), (5000,)] * 20  # > 64 tensors: several launches
    ps = [torch.randn(*s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().to(dev).requires_grad_(True) for p in ps]
    o_ref, o_mine = torch.optim.Adam(ref, lr=1e-2), FusedAdam(mine, lr=1e-2)
    for step in range(3):
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad, m.grad = gr, gr.to(dev)
        o_ref.step()
        o_mine.step()
    for r, m in zip(ref, mine):
        torch.testing.assert_close(m.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,t,d,cs", [(2, 128, 16, (64,)), (1, 256, 32, (128,)), (2, 1024, 32, (128, 128)), (1, 512, 64, (256,)), (1, 384, 64, (64, 64))])
def test_fused_attention_forward(dev, FF, n, t, d, cs):
    """flash-style fused kernel against fp64 softmax(q q^T) v, incl. a query whose maximum jumps late (forces the lazy
    rescale branch) and large logits"""
    g = torch.Generator().manual_seed(t + d)
    q = torch.randn(n, t, d, generator=g) * 0.8
    q[0, 5] *= 6.0          # a key with a huge norm: every query's running maximum jumps when this tile arrives
    q[0, t - 3] *= 9.0      # ... and again in the last tile
    vs = [torch.randn(n, t, c, generator=g) for c in cs]
    att = torch.softmax(q.double() @ q.double().transpose(1, 2), -1)
    qd = q.to(dev)
    vds = [v.to(dev) for v in vs]
    assert FF._fused_attn_ok(qd, vds)
    res = FF.self_attention(qd, vds)
    for r, v in zip(res, vs):
        torch.testing.assert_close(r.cpu().double(), att @ v.double(), rtol=1e-4, atol=2e-5)
    # and the unfused composition agrees with the fused kernel
    FF.FUSED_ATTENTION = False
    try:
        res2 = FF.self_attention(qd, vds)
    finally:
        FF.FUSED_ATTENTION = True
    for a, b in zip(res, res2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5)


def test_ssim_metric(dev, golden):
    """modules/evaluations/ssim.py mirror against the reference's own outputs"""
    from face_mask_inpaint_amd.modules.evaluations.ssim import SSIM, ssim

    fx = golden("ssim.pt")
    a, b = fx["a"].to(dev), fx["b"].to(dev)
    torch.testing.assert_close(ssim(a, b).cpu(), fx["mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ssim(a, b, size_average=False).cpu(), fx["per_image"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(SSIM()(a, a).cpu(), fx["same"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# thin-output 3x3 convolution (K <= 4): the generator's Output block, base_function.py:367-398
@pytest.mark.parametrize("n,c,k,h,w,pad_mode,act", [(2, 32, 3, 20, 24, 1, 1), (1, 32, 3, 64, 64, 1, 1), (2, 8, 1, 9, 7, 0, 0), (2, 4, 4, 5, 6, 1, 0),
                                                    (1, 64, 2, 12, 12, 0, 2), (3, 16, 3, 3, 3, 1, 0), (1, 32, 3, 130, 70, 0, 0)])
def test_thin_output_conv(dev, FF, n, c, k, h, w, pad_mode, act):
    from face_mask_inpaint_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(h * 10 + c + k)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = (torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(k, generator=g, requires_grad=True)
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect") if pad_mode else F.pad(x, (1, 1, 1, 1))
    pre = F.conv2d(xp, wt_, b)
    y = torch.tanh(pre) if act == 1 else (torch.relu(pre) if act == 2 else pre)
    gpre = torch.randn(pre.shape, generator=g)  # cotangent of the pre-activation (the activation backward is a separate kernel)
    pre.backward(gpre)
    d, oh, ow = FF.conv_desc(n, h, w, c, k, 3, 3, 1, 1, pad_mode)
    assert lib.conv2d_thin_supported(C.byref(d)) == 1
    wf, wtp = [t.to(dev) for t in pack(wt_.detach())]
    xh, gh, bd = nhwc(x.detach()).to(dev), nhwc(gpre).to(dev), b.detach().to(dev)
    st = FF._st()
    out = torch.full((n, h, w, k), float("nan"), device=dev)
    lib.conv2d_fwd_f32(C.byref(d), FF._p(xh), FF._p(wf), FF._p(bd), None, FF._p(out), act, 1, 0, st)  # routes to the thin kernel
    torch.testing.assert_close(out.cpu(), nhwc(y.detach()), rtol=1e-5, atol=1e-5)
    res = torch.randn(n, h, w, k, generator=g).to(dev)
    out2 = torch.empty_like(out)
    lib.conv2d_thin_fwd_f32(C.byref(d), FF._p(xh), FF._p(wf), None, FF._p(res), FF._p(out2), 0, st)
    torch.testing.assert_close(out2.cpu(), nhwc((pre - b.view(1, -1, 1, 1)).detach()) + res.cpu(), rtol=1e-5, atol=1e-5)
    dx = torch.full((n, h, w, c), float("nan"), device=dev)
    lib.conv2d_thin_dgrad_f32(C.byref(d), FF._p(gh), FF._p(wtp), FF._p(dx), st)
    torch.testing.assert_close(dx.cpu(), nhwc(x.grad), rtol=1e-5, atol=1e-5)
    dwf, db = torch.zeros_like(wf), torch.zeros(k, device=dev)
    lib.conv2d_wgrad_f32(C.byref(d), FF._p(xh), FF._p(gh), FF._p(dwf), FF._p(db), 1, 0, st)  # routes to the thin kernel
    tol = 1e-4 * max(1.0, (n * h * w / 2048.0) ** 0.5)
    torch.testing.assert_close(dwf.cpu(), pack(wt_.grad)[0], rtol=1e-4, atol=tol)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=tol)
    if pad_mode == 0:  # the generic entry point routes zero-padded thin adjoints too
        dx2 = torch.full((n, h, w, c), float("nan"), device=dev)
        lib.conv2d_dgrad_f32(C.byref(d), FF._p(gh), FF._p(wtp), None, None, FF._p(dx2), 1, 0, st)
        torch.testing.assert_close(dx2.cpu(), nhwc(x.grad), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,c,k,h,w,ksz,stride,slope", [(2, 32, 64, 33, 31, 3, 1, 0.1), (2, 128, 128, 16, 16, 3, 1, 0.1), (1, 16, 48, 20, 20, 1, 1, 0.2),
                                                         (2, 32, 64, 32, 32, 3, 2, 0.1), (2, 64, 64, 24, 24, 3, 1, 0.0), (4, 128, 128, 8, 8, 3, 1, 0.1)])
def test_conv_with_input_activation(dev, FF, n, c, k, h, w, ksz, stride, slope):
    """FF.conv2d(in_act=("apply", s)) = conv(lrelu(x, s)) and in_act=("mask", s) on an already activated input: values and gradients
    against autograd; the input gradient is produced by fmi_conv2d_dgrad_masked_f32 (activation derivative in the adjoint's epilogue,
    also when the reduction is split over workgroups)"""
    from face_mask_inpaint_amd.functional import PackedWeight

    g = torch.Generator().manual_seed(h * 11 + c + k)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = (torch.randn(k, c, ksz, ksz, generator=g) / (c * ksz * ksz) ** 0.5).requires_grad_(True)
    b = torch.randn(k, generator=g, requires_grad=True)
    pad = ksz // 2
    a = F.leaky_relu(x, slope)
    y = F.conv2d(a, wt_, b, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    wf, wtp = [t.to(dev) for t in pack(wt_.detach())]
    for mode in ("apply", "mask"):
        wfd = wf.clone().requires_grad_(True)
        xin = nhwc(x.detach() if mode == "apply" else a.detach()).to(dev).requires_grad_(True)
        bd = b.detach().to(dev).requires_grad_(True)
        out = FF.conv2d(xin, PackedWeight(wfd, wtp, k, c, ksz, ksz), bd, None, stride, pad, 0, FF.ACT_NONE, (mode, slope))
        torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-4, atol=1e-5)
        out.backward(nhwc(gy).to(dev))
        torch.testing.assert_close(xin.grad.cpu(), nhwc(x.grad), rtol=1e-4, atol=1e-5)
        tol = 2e-4 * max(1.0, (n * y.shape[2] * y.shape[3] / 2048.0) ** 0.5)
        torch.testing.assert_close(wfd.grad.cpu(), pack(wt_.grad)[0], rtol=1e-4, atol=tol)
        torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=tol)


@pytest.mark.parametrize("n,c,k,h,w,pad_mode", [(2, 32, 3, 20, 24, 1), (1, 32, 3, 70, 33, 1), (2, 32, 4, 9, 12, 0), (1, 16, 3, 12, 12, 1)])
def test_output_block_fused_lrelu_conv_tanh(dev, FF, n, c, k, h, w, pad_mode):
    """FF.lrelu_conv2d = tanh(conv3x3(pad(lrelu(x)))) (base_function.py:386-396) with the LeakyReLU folded into the thin-output
    kernels (C = 32) or, for other widths, the LeakyReLU + convolution pair: values and all gradients against autograd"""
    from face_mask_inpaint_amd.functional import PackedWeight

    g = torch.Generator().manual_seed(h * 13 + c + k)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = (torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(k, generator=g, requires_grad=True)
    a = F.leaky_relu(x, 0.1)
    ap = F.pad(a, (1, 1, 1, 1), mode="reflect") if pad_mode else F.pad(a, (1, 1, 1, 1))
    y = torch.tanh(F.conv2d(ap, wt_, b))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    wf, wtp = [t.to(dev) for t in pack(wt_.detach())]
    wf.requires_grad_(True)
    xd, bd = nhwc(x.detach()).to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    out = FF.lrelu_conv2d(xd, PackedWeight(wf, wtp, k, c, 3, 3), bd, 0.1, 1, pad_mode, FF.ACT_TANH)
    torch.testing.assert_close(out.detach().cpu(), nhwc(y.detach()), rtol=1e-5, atol=1e-5)
    out.backward(nhwc(gy).to(dev))
    torch.testing.assert_close(xd.grad.cpu(), nhwc(x.grad), rtol=1e-4, atol=1e-5)
    tol = 1e-4 * max(1.0, (n * h * w / 2048.0) ** 0.5)
    torch.testing.assert_close(wf.grad.cpu(), pack(wt_.grad)[0], rtol=1e-4, atol=tol)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=tol)


@pytest.mark.parametrize("n,c,k,h,w", [(2, 3, 64, 40, 36), (1, 3, 32, 130, 200), (2, 1, 16, 9, 7), (1, 4, 8, 3, 3), (2, 2, 4, 17, 5)])
def test_thin_input_conv_adjoint(dev, FF, n, c, k, h, w):
    """input gradient of a thin-INPUT 3x3 convolution (VGG16's first layer 3 -> 64, loss.py:45-65): the dedicated entry and the
    generic dgrad entry that routes to it, against autograd"""
    from face_mask_inpaint_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(h * 7 + c + k)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt_ = torch.randn(k, c, 3, 3, generator=g) / (c * 9) ** 0.5
    gy = torch.randn(n, k, h, w, generator=g)
    F.conv2d(x, wt_, padding=1).backward(gy)
    d, _, _ = FF.conv_desc(n, h, w, c, k, 3, 3, 1, 1, 0)
    wtp = pack(wt_)[1].to(dev)
    gh, st = nhwc(gy).to(dev), FF._st()
    for fn in (lambda dx: lib.conv2d_thin_input_dgrad_f32(C.byref(d), FF._p(gh), FF._p(wtp), FF._p(dx), st),
               lambda dx: lib.conv2d_dgrad_f32(C.byref(d), FF._p(gh), FF._p(wtp), None, None, FF._p(dx), 1, 0, st)):
        dx = torch.full((n, h, w, c), float("nan"), device=dev)
        fn(dx)
        torch.testing.assert_close(dx.cpu(), nhwc(x.grad), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("n,h,c", [(3, 16, 64), (2, 64, 128), (16, 32, 512), (2, 8, 256), (2, 16, 12)])
def test_global_average_pool(dev, FF, n, h, c):
    """AdaptiveAvgPool2d(1) of the SE modules (helpers.py:56-72): reduction path (and the windowed fallback for C = 12)"""
    g = torch.Generator().manual_seed(n * 100 + c)
    x = torch.randn(n, h, h, c, generator=g)
    xd = x.to(dev).requires_grad_(True)
    y = FF.avg_pool(xd, h)
    assert y.shape == (n, 1, 1, c)
    torch.testing.assert_close(y.detach().cpu().view(n, c), x.mean(dim=(1, 2)), rtol=1e-5, atol=1e-6)
    gy = torch.randn(n, 1, 1, c, generator=g)
    y.backward(gy.to(dev))
    torch.testing.assert_close(xd.grad.cpu(), (gy / (h * h)).expand(n, h, h, c), rtol=1e-6, atol=1e-8)


def test_ranger_against_reference(dev, golden):
    """Ranger (RAdam + Lookahead + gradient centralisation, modules/psp/ranger.py) as one multi-tensor launch pair: 13 steps on the
    gradients of tests/golden/ranger.pt against the parameters the reference's optimiser produced (crosses the rectification
    threshold and two Lookahead syncs)"""
    from face_mask_inpaint_amd.modules.psp.ranger import Ranger

    fx = golden("ranger.pt")
    ps = [torch.nn.Parameter(p.clone().to(dev)) for p in fx["p0"]]
    opt = Ranger(ps, **fx["cfg"])
    for step, (gs, want) in enumerate(zip(fx["grads"], fx["params"])):
        for p, g in zip(ps, gs):
            p.grad = g.clone().to(dev)
        opt.step()
        for i, (p, w) in enumerate(zip(ps, want)):
            torch.testing.assert_close(p.detach().cpu(), w, rtol=2e-5, atol=2e-6, msg=lambda m, i=i, step=step: f"step {step} tensor {i}: {m}")
    assert set(opt.state[ps[0]]) == {"step", "exp_avg", "exp_avg_sq", "slow_buffer"} and opt.state[ps[0]]["step"] == 13
    with pytest.raises(ValueError):
        Ranger(ps, alpha=1.5)


def _pieces_to_float(p3, taps, cred, nout):
    """[3][taps][cred/8][nout][8] bf16 piece images -> three fp32 tensors [taps][cred][nout]"""
    return [p3.view(3, taps, cred // 8, nout, 8)[i].permute(0, 1, 3, 2).reshape(taps, cred, nout).float() for i in range(3)]


@pytest.mark.parametrize("n,c,k,h,w,ksz,stride,pad", [(2, 32, 64, 33, 31, 3, 1, 1), (2, 128, 128, 16, 16, 3, 1, 1), (1, 64, 256, 20, 20, 3, 1, 1), (2, 16, 48, 17, 15, 1, 1, 0),
                                                      (3, 64, 32, 20, 20, 3, 2, 1), (2, 32, 64, 16, 16, 4, 2, 1), (2, 48, 80, 9, 11, 5, 1, 2),
                                                      # deep reductions on small maps: the piece images under a split reduction (tap-reuse kernel, strided adjoint phases)
                                                      (4, 512, 256, 8, 8, 3, 1, 1), (4, 512, 256, 8, 8, 3, 2, 1), (8, 512, 512, 2, 2, 3, 2, 1)])
def test_conv_weight_piece_images(dev, FF, n, c, k, h, w, ksz, stride, pad):
    """fmi_weight_prepare_f32 also writes both packs as three bf16 piece images (fmi_conv_desc.w3): the pieces add up to the fp32 pack
    EXACTLY (bf16x6 products lose nothing of the operands), sit in the [piece][tap][Cred/8][Nout][8] layout, and a convolution /
    adjoint / ConvTranspose given the pieces returns what it returns when it splits the weight fragments itself."""
    from face_mask_inpaint_amd import _lib

    lib = _lib.lib()
    g = torch.Generator().manual_seed(c * 7 + k)
    wt_ = (torch.randn(k, c, ksz, ksz, generator=g) / (c * ksz * ksz) ** 0.5).to(dev)
    wt_[0, 0, 0, 0] = 1.0e-20  # a tiny and a large value go through the split as well (exactness needs |x| >= 2^-100: the third piece is 2^-16 |x|)
    wt_[1, 0, 0, 0] = 1.5e30
    (pw,) = FF.prepare_weights([(wt_, None, None)])
    taps = ksz * ksz
    wf3, wt3 = pw.w3
    assert (wf3 is not None) == (c % 16 == 0) and (wt3 is not None) == (k % 16 == 0)
    for p3, pack_, cred, nout in ((wf3, pw.wf, c, k), (wt3, pw.wt, k, c)):
        if p3 is None:
            continue
        x0, x1, x2 = _pieces_to_float(p3, taps, cred, nout)
        assert torch.equal((x0.double() + x1.double() + x2.double()).float(), pack_.detach()), "pieces do not add up to the pack"
        big = pack_.detach().abs() > 1e-30
        assert float((x1.abs() / pack_.detach().abs().clamp_min(1e-38))[big].max()) <= 2.0 ** -8 and float((x2.abs() / pack_.detach().abs().clamp_min(1e-38))[big].max()) <= 2.0 ** -16
    st = FF._st()
    x = torch.randn(n, h, w, c, generator=g).to(dev)
    outs = []
    for w3 in (None, wf3):
        d, oh, ow = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, w3=w3)
        y = torch.full((n, oh, ow, k), float("nan"), device=dev)
        lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(pw.wf.detach()), None, None, FF._p(y), 0, 1, 0, st)
        outs.append(y)
    assert torch.isfinite(outs[0][..., 2:]).all()
    torch.testing.assert_close(outs[1][..., 2:], outs[0][..., 2:], rtol=2e-6, atol=2e-6 * float(outs[0][..., 2:].abs().max()))
    gy = torch.randn(n, oh, ow, k, generator=g).to(dev)
    gy[..., :2] = 0  # columns 0 / 1 carry the 1e-20 / 1.5e30 weights
    outs = []
    for w3 in (None, wt3):
        d, _, _ = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, w3=w3)
        dx = torch.full((n, h, w, c), float("nan"), device=dev)
        lib.conv2d_dgrad_f32(C.byref(d), FF._p(gy), FF._p(pw.wt), None, None, FF._p(dx), 1, 0, st)
        outs.append(dx)
    torch.testing.assert_close(outs[1], outs[0], rtol=2e-6, atol=2e-6 * float(outs[0].abs().max()))


def test_bf16x6_product_accuracy(dev, FF):
    """the fp32 GEMM on the bf16 matrix pipe (csrc/x6.h) against float64: error per element below 1e-6 of sum |a||b|, on average below
    5e-8 (rocBLAS' fp32 GEMM on the same operands: 7e-7 / 3.4e-8; measured here 4.4e-7 / 2.9e-8), with rows spanning 1e-6 .. 1e6, and an
    identity operand passing the other one through exactly"""
    torch.manual_seed(0)
    a = torch.randn(512, 1024, device=dev) * torch.exp(torch.randn(512, 1, device=dev) * 4)
    b = torch.randn(1024, 384, device=dev) * torch.exp(torch.randn(1024, 1, device=dev))
    c = torch.empty(512, 384, device=dev)
    FF.gemm_raw(FF._p(a), FF._p(b), FF._p(c), 512, 384, 1024, (1024, 1), (384, 1), (384, 1))
    ref = a.double() @ b.double()
    den = a.double().abs() @ b.double().abs()
    err = ((c.double() - ref).abs() / den)
    assert float(err.max()) <= 1e-6 and float(err.mean()) <= 5e-8, (float(err.max()), float(err.mean()))
    eye = torch.eye(256, device=dev)
    m = torch.randn(256, 256, device=dev) * 1e3
    out = torch.empty(256, 256, device=dev)
    FF.gemm_raw(FF._p(eye), FF._p(m), FF._p(out), 256, 256, 256, (256, 1), (256, 1), (256, 1))
    assert torch.equal(out, m)  # x = x0 + x1 + x2 exactly, and 1 * piece is exact in the fp32 accumulator


@pytest.mark.parametrize("n,t,d,cs", [(8, 2048, 64, (256,)), (8, 2048, 32, (128, 128)), (8, 2048, 32, (128,)), (8, 2048, 64, (64, 64)), (4, 4096, 64, (128, 128))])
def test_fused_attention_backward_key_block_structure(dev, FF, n, t, d, cs):
    """sizes at which fmi_attention_bwd_f32 takes the second structure ((T / 128) * N >= 128: attn_bwd2_x6_kernel, every instantiation):
    dQ (query side + key side, fp32 atomics) and dV against float64 autograd of softmax(q q^T) v on the same device"""
    g = torch.Generator().manual_seed(t + d + sum(cs))
    q = (torch.randn(n, t, d, generator=g) * 0.5).to(dev)
    vs = [torch.randn(n, t, c, generator=g).to(dev) for c in cs]
    gos = [torch.randn(n, t, c, generator=g).to(dev) for c in cs]
    q64 = q.double().requires_grad_(True)
    v64 = [v.double().requires_grad_(True) for v in vs]
    att = torch.softmax(q64 @ q64.transpose(1, 2), -1)
    outs = [att @ v for v in v64]
    torch.autograd.backward(outs, [go.double() for go in gos])
    del att
    qd = q.clone().requires_grad_(True)
    vds = [v.clone().requires_grad_(True) for v in vs]
    res = FF.self_attention(qd, vds)
    for r, o in zip(res, outs):
        torch.testing.assert_close(r.detach().double(), o.detach(), rtol=1e-4, atol=1e-5)
    torch.autograd.backward(res, gos)
    gq, gq64 = qd.grad.double(), q64.grad
    scale = float(gq64.abs().max())
    assert float((gq - gq64).abs().max()) <= 2e-5 * scale + 1e-6, (float((gq - gq64).abs().max()), scale)  # measured ~3e-6 of the largest entry
    for vd, v in zip(vds, v64):
        torch.testing.assert_close(vd.grad.double(), v.grad, rtol=1e-4, atol=2e-5)


def test_adam_guard_skips_a_non_finite_step(dev, FF):
    """FusedAdam(capturable=True).step(guard=loss): a NaN / inf device scalar turns the step into a no-op -- parameters, both moments and
    the device step count unchanged (train_psp.py:328-331 as a device-side predicate); a finite guard equals the unguarded step"""
    from face_mask_inpaint_amd.optim import FusedAdam

    torch.manual_seed(0)
    p0 = torch.randn(1000, device=dev)
    g = torch.randn(1000, device=dev)
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa, ob = FusedAdam([pa], lr=1e-2, capturable=True), FusedAdam([pb], lr=1e-2, capturable=True)
    for guard in (torch.tensor([1.5], device=dev), torch.tensor([float("nan")], device=dev), torch.tensor([float("inf")], device=dev),
                  torch.tensor([0.25], device=dev)):
        pa.grad, pb.grad = g.clone(), g.clone()
        before = (pa.detach().clone(), int(oa.param_groups[0].get("step_dev", torch.zeros(1)).item()))
        oa.step(guard=guard)
        if torch.isfinite(guard).item():
            ob.step()
            assert torch.equal(pa.detach(), pb.detach())
        else:
            assert torch.equal(pa.detach(), before[0]) and int(oa.param_groups[0]["step_dev"].item()) == before[1]
    assert int(oa.param_groups[0]["step_dev"].item()) == 2 == int(ob.param_groups[0]["step_dev"].item())
    assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"])


def test_fused_adam_state_dict_crosses_modes(dev, FF):
    """state written by the capturable path (device counter) loads into the host-counter path and back, through a CPU round trip, and the
    continued trajectories equal an uninterrupted torch.optim.Adam run (<= 1e-6: same update arithmetic)"""
    from face_mask_inpaint_amd.optim import FusedAdam

    torch.manual_seed(1)
    w0 = torch.randn(257, device=dev)
    grads = [torch.randn(257, device=dev) for _ in range(6)]
    ref = torch.nn.Parameter(w0.clone())
    oref = torch.optim.Adam([ref], lr=1e-2)
    for g in grads:
        ref.grad = g.clone()
        oref.step()
    p = torch.nn.Parameter(w0.clone())
    o = FusedAdam([p], lr=1e-2, capturable=True)
    for g in grads[:2]:
        p.grad = g.clone()
        o.step()
    import io

    buf = io.BytesIO()
    torch.save(o.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu", weights_only=True)  # a checkpoint read back on the CPU: the counter arrives as a CPU tensor
    o2 = FusedAdam([p], lr=1e-2, capturable=False)
    o2.load_state_dict(sd)
    for g in grads[2:4]:
        p.grad = g.clone()
        o2.step()
    assert o2.state[p]["step"] == 4
    o3 = FusedAdam([p], lr=1e-2, capturable=True)
    o3.load_state_dict(o2.state_dict())
    for g in grads[4:]:
        p.grad = g.clone()
        o3.step()
    assert int(o3.param_groups[0]["step_dev"].item()) == 6 and o3.param_groups[0]["step_dev"].is_cuda
    torch.testing.assert_close(p.detach(), ref.detach(), rtol=1e-6, atol=1e-6)


def test_weight_pieces_on_the_register_staged_kernel():
    """FMI_DMA_OFF=2 (debug switch of tools/bench_tools/bisect.py, trace.py) sends convolutions to the register-staged kernel; with piece
    images of the weights given (fmi_conv_desc.w3) that kernel now rebuilds the fp32 weight from the pieces -- round 2 read zeros there.
    The switch is read once per process, so this runs in a child process."""
    import os
    import subprocess
    import sys

    code = r"""
import ctypes as C, torch
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0"); lib = _lib.lib(); st = FF._st()
g = torch.Generator().manual_seed(3)
w = (torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(dev)
(pw,) = FF.prepare_weights([(w, None, None)])
x = torch.randn(2, 20, 20, 32, generator=g).to(dev)
outs = []
for w3 in (None, pw.w3[0]):
    d, oh, ow = FF.conv_desc(2, 20, 20, 32, 32, 3, 3, 2, 1, w3=w3)   # stride 2: the generic loader pair
    y = torch.full((2, oh, ow, 32), float("nan"), device=dev)
    lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(pw.wf.detach()), None, None, FF._p(y), 0, 1, 0, st)
    outs.append(y)
assert float(outs[0].abs().max()) > 0.1
assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[0].abs().max()), float((outs[0] - outs[1]).abs().max())
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FMI_DMA_OFF="2", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_bf16x6_edge_values_on_the_matrix_pipe(dev, FF):
    """what csrc/x6.h documents, pinned on the MFMA path: (a) operands down to 2^-100 keep full fp32 accuracy (the third piece is still a
    normal bf16 number); below that pieces become bf16 SUBNORMALS, which the matrix pipe flushes to zero: at 2^-120 the second and third piece
    are gone and the result is the bf16-rounded operand (measured 1.2e-4, bound 2^-8 relative, no NaN) -- magnitudes (1e-36) no activation,
    weight or gradient of this path comes near; (b) an infinite
    operand gives NaN (inf - inf inside the split) where an fp32 product gives inf; (c) NaN stays NaN; (d) zeros and signed zeros give 0"""
    n = 64
    eye = torch.eye(n, device=dev)

    def through(m):
        out = torch.empty(n, n, device=dev)
        FF.gemm_raw(FF._p(eye), FF._p(m), FF._p(out), n, n, n, (n, 1), (n, 1), (n, 1))
        return out

    torch.manual_seed(0)
    small = (torch.rand(n, n, device=dev) + 0.5) * 2.0 ** -100
    assert torch.equal(through(small), small)
    tiny = (torch.rand(n, n, device=dev) + 0.5) * 2.0 ** -120
    t = through(tiny)
    assert torch.isfinite(t).all() and float(((t - tiny).abs() / tiny).max()) <= 2.0 ** -8
    m = torch.randn(n, n, device=dev)
    m[3, 5] = float("inf")
    m[7, 9] = float("nan")
    m[0, 0], m[1, 1] = 0.0, -0.0
    o = through(m)
    assert torch.isnan(o[3, 5]) and torch.isnan(o[7, 9]) and float(o[0, 0]) == 0.0 and float(o[1, 1]) == 0.0
    keep = torch.ones(n, n, dtype=torch.bool, device=dev)
    keep[3, :] = False  # the identity row that multiplies the inf spreads NaN along its output row only
    keep[7, :] = False
    keep[:, 5] = False
    keep[:, 9] = False
    assert torch.equal(o[keep], m[keep])
