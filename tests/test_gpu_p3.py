"""GPU: bf16 piece images of ACTIVATIONS (csrc/conv_p3.h, csrc/p3.hip) through the C ABI.

The pieces are an exact restatement of the fp32 tensor, so the convolutions that read them (fmi_conv_desc.x3) must return what the
in-wave split returns: the same exact bf16 products summed in fp32, in an order that may differ -> 2e-6 of the largest entry.  The
reference for the values themselves stays torch fp32 on the CPU (base_function.py:207-364: Conv2d / ConvTranspose2d of the ResBlocks)."""
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


def _lib():
    from face_mask_inpaint_amd import _lib

    return _lib.lib()


def _split(FF, x, op=0, slope=0.0, want_y=False):
    c = x.shape[-1]
    x3 = torch.full((x.numel() * 3,), float("nan"), device=x.device, dtype=torch.bfloat16)
    y = torch.full_like(x, float("nan")) if want_y else None
    _lib().split3_f32(FF._p(x), C.c_void_p(x3.data_ptr()), FF._p(y), x.numel() // c, c, op, slope, FF._st())
    return x3, y


def _pieces(x3, shape):
    c = shape[-1]
    v = x3.view(-1, c // 16, 3, 16).float()  # [pixel][group][piece][16]
    return [v[:, :, i, :].reshape(shape) for i in range(3)]


@pytest.mark.parametrize("shape", [(2, 5, 7, 16), (1, 33, 9, 48), (3, 16, 16, 256)])
def test_split3_pieces_are_exact(dev, FF, shape):
    """x = x0 + x1 + x2 bit for bit, x0 = round-to-nearest bf16 of x, |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|; the lrelu form cuts lrelu(x) and
    hands the fp32 value back; the inverse (fmi_merge3_f32) restores the tensor"""
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(shape, generator=g) * torch.exp(torch.randn(shape, generator=g) * 3)).to(dev)
    x.view(-1)[:4] = torch.tensor([0.0, -0.0, 1.0e-20, -3.0e30], device=dev)
    x3, _ = _split(FF, x)
    x0, x1, x2 = _pieces(x3, shape)
    assert torch.equal((x0.double() + x1.double() + x2.double()).float(), x)
    assert torch.equal(x0, x.bfloat16().float())
    big = x.abs() > 1e-30
    assert float((x1.abs() / x.abs().clamp_min(1e-38))[big].max()) <= 2.0 ** -8
    assert float((x2.abs() / x.abs().clamp_min(1e-38))[big].max()) <= 2.0 ** -16
    back = torch.full_like(x, float("nan"))
    _lib().merge3_f32(C.c_void_p(x3.data_ptr()), FF._p(back), x.numel() // shape[-1], shape[-1], FF._st())
    assert torch.equal(back, x)
    y3, y = _split(FF, x, op=1, slope=0.1, want_y=True)
    ref = torch.where(x > 0, x, x * 0.1)
    assert torch.equal(y, ref)
    y0, y1, y2 = _pieces(y3, shape)
    assert torch.equal((y0.double() + y1.double() + y2.double()).float(), ref)


@pytest.mark.parametrize("pixels,c", [(4096, 64), (777, 32), (33, 512), (5000, 16), (300, 256)])
def test_split3_with_bias_gradient(dev, FF, pixels, c):
    """fmi_split3_colsum_f32: the piece image is the one fmi_split3_f32 writes, bit for bit, and colsum += the sum over the pixels (fp32 partial
    sums per thread and workgroup: 2e-6 of sum |dy| against float64); shapes a thread cannot keep one channel chunk for are refused; and Conv2d's
    backward takes the bias gradient from it (against fmi_bias_grad_f32 through the reproducible mode)"""
    g = torch.Generator().manual_seed(pixels + c)
    x = torch.randn(pixels, c, generator=g).to(dev)
    ref3, _ = _split(FF, x)
    x3 = torch.zeros_like(ref3)
    cs = torch.full((c,), 1.0, device=dev)
    _lib().split3_colsum_f32(FF._p(x), C.c_void_p(x3.data_ptr()), FF._p(cs), pixels, c, FF._st())
    assert torch.equal(x3, ref3)
    want = x.double().sum(0) + 1.0
    assert float((cs.double() - want).abs().max()) <= 2e-6 * float(x.double().abs().sum(0).max())


def test_split3_with_bias_gradient_refusals_and_autograd(dev, FF):
    from face_mask_inpaint_amd._lib import FmiError

    x = torch.zeros(8, 48, device=dev)
    x3 = torch.zeros(8 * 48 * 3, device=dev, dtype=torch.bfloat16)
    with pytest.raises(FmiError):  # 256 % (48 / 8) != 0
        _lib().split3_colsum_f32(FF._p(x), C.c_void_p(x3.data_ptr()), FF._p(torch.zeros(48, device=dev)), 8, 48, FF._st())
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(2, 96, 96, 64, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = torch.zeros(128, device=dev, requires_grad=True)
    up = torch.randn(2, 96, 96, 128, generator=g).to(dev)
    grads = {}
    for det in (False, True):
        with FF.deterministic(det):
            for t in (xin, w, b):
                t.grad = None
            (pw,) = FF.prepare_weights([(w, None, None)])
            (FF.conv2d(xin, pw, b, pad=1) * up).sum().backward()
            grads[det] = b.grad.clone()
    want = up.double().sum((0, 1, 2))
    for det in (False, True):
        assert float((grads[det].double() - want).abs().max()) <= 2e-6 * float(up.double().abs().sum((0, 1, 2)).max())


def test_split3_argument_checks(dev, FF):
    x = torch.zeros(4, 24, device=dev)
    x3 = torch.zeros(4 * 24 * 3, device=dev, dtype=torch.bfloat16)
    from face_mask_inpaint_amd._lib import FmiError

    with pytest.raises(FmiError):  # C % 16 != 0
        _lib().split3_f32(FF._p(x), C.c_void_p(x3.data_ptr()), None, 4, 24, 0, 0.0, FF._st())


CASES = [
    # n, c, k, h, w, ksz, stride, pad
    (2, 64, 64, 40, 36, 3, 1, 1),     # tap-reuse kernel, ragged rows (2880 pixels: 256-row tiles with a tail)
    (1, 128, 192, 33, 31, 3, 1, 1),   # Nout not a multiple of the 128-column tile
    (2, 256, 128, 16, 16, 3, 1, 1),   # small map: split reduction (atomic epilogue)
    (2, 64, 128, 24, 24, 1, 1, 0),    # 1x1: generic piece kernel
    (2, 64, 64, 32, 32, 3, 2, 1),     # stride 2: generic piece kernel; its adjoint = ConvTranspose2d forward by sub-pixel phases
    (1, 32, 64, 20, 20, 3, 1, 1),     # 32 reduction channels (two k-steps per tap)
    (1, 64, 32, 64, 64, 3, 1, 1),     # 32 output columns
    (8, 128, 128, 64, 64, 3, 1, 1),   # 32768 pixels: the 256 x 128 eight-wave tile
]


@pytest.mark.parametrize("n,c,k,h,w,ksz,stride,pad", CASES)
def test_conv_with_activation_pieces(dev, FF, n, c, k, h, w, ksz, stride, pad):
    """forward and adjoint given fmi_conv_desc.x3 against (a) the same entry without it (<= 2e-6 of the largest entry: same exact products,
    another summation order) and (b) torch fp32 on the CPU (1e-5 relative, the bound of tests/test_gpu_kernels.py)"""
    lib = _lib()
    st = FF._st()
    g = torch.Generator().manual_seed(c * 3 + k + h)
    wt_ = (torch.randn(k, c, ksz, ksz, generator=g) / (c * ksz * ksz) ** 0.5)
    (pw,) = FF.prepare_weights([(wt_.to(dev), None, None)])
    wf3, wt3 = pw.w3
    x = torch.randn(n, h, w, c, generator=g)
    xd = x.to(dev)
    x3, _ = _split(FF, xd)
    bias = torch.randn(k, generator=g).to(dev)
    outs = []
    for xx3 in (None, x3):
        d, oh, ow = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, w3=wf3, x3=xx3)
        y = torch.full((n, oh, ow, k), float("nan"), device=dev)
        lib.conv2d_fwd_f32(C.byref(d), FF._p(xd), FF._p(pw.wf.detach()), FF._p(bias), None, FF._p(y), 0, 1, 0, st)
        outs.append(y)
    scale = float(outs[0].abs().max())
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=2e-6 * scale)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt_, bias.cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    torch.testing.assert_close(outs[1].cpu(), ref, rtol=1e-5, atol=1e-5 * scale)
    # piece image of the result on request (fmi_conv_desc.y3)
    if k % 16 == 0:
        d, oh, ow = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, w3=wf3, x3=x3)
        y3 = torch.full((n * oh * ow * k * 3,), float("nan"), device=dev, dtype=torch.bfloat16)
        d.y3 = y3.data_ptr()
        y = torch.full((n, oh, ow, k), float("nan"), device=dev)
        lib.conv2d_fwd_f32(C.byref(d), FF._p(xd), FF._p(pw.wf.detach()), FF._p(bias), None, FF._p(y), 0, 1, 0, st)
        y0, y1, y2 = _pieces(y3, (n, oh, ow, k))
        assert torch.equal((y0.double() + y1.double() + y2.double()).float(), y)
    # adjoint: dy pieces as the activation operand, wt pieces as the weights
    gy = torch.randn(n, oh, ow, k, generator=g)
    gyd = gy.to(dev)
    gy3, _ = _split(FF, gyd)
    outs = []
    for gg3 in (None, gy3):
        d, _, _ = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad, w3=wt3, x3=gg3)
        dx = torch.full((n, h, w, c), float("nan"), device=dev)
        lib.conv2d_dgrad_f32(C.byref(d), FF._p(gyd), FF._p(pw.wt), None, None, FF._p(dx), 1, 0, st)
        outs.append(dx)
    scale = float(outs[0].abs().max())
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=2e-6 * scale)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.conv2d(xr, wt_, None, stride=stride, padding=pad).backward(gy.permute(0, 3, 1, 2))
    torch.testing.assert_close(outs[1].cpu(), xr.grad.permute(0, 2, 3, 1), rtol=1e-5, atol=1e-5 * scale)


def test_autograd_path_takes_the_pieces(dev, FF, monkeypatch):
    """FF.conv2d / FF.conv_transpose2d with the piece path forced on (every size) against the same calls with it off: outputs and input
    gradients within 2e-6 of their largest entry, weight gradients unchanged (they do not read the pieces yet)"""
    g = torch.Generator().manual_seed(5)
    w1 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    w2 = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)  # ConvTranspose2d weight [in][out][kh][kw] -> conv view rows = in
    x = torch.randn(2, 24, 24, 64, generator=g).to(dev).requires_grad_(True)
    res = {}
    for on in (False, True):
        monkeypatch.setattr(FF, "P3_ENABLED", on)
        monkeypatch.setattr(FF, "P3_MIN_PIXELS", 0)
        monkeypatch.setattr(FF, "P3_MIN_WORK", 0)
        for t in (w1, w2, x):
            t.grad = None
        pw1, pw2 = FF.prepare_weights([(w1, None, None), (w2, None, None)])
        h = FF.conv2d(x, pw1, None, None, 1, 1, in_act=("apply", 0.1))
        y = FF.conv_transpose2d(h, pw2)
        (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
        res[on] = (y.detach().clone(), x.grad.clone(), w1.grad.clone(), w2.grad.clone())
    for a, b in zip(res[True][:2], res[False][:2]):
        torch.testing.assert_close(a, b, rtol=0, atol=2e-6 * float(b.abs().max()))
    for a, b in zip(res[True][2:], res[False][2:]):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-5 * float(b.abs().max()))


@pytest.mark.parametrize("n,c,k,h,w,ksz,stride,pad", [(2, 64, 64, 40, 36, 3, 1, 1), (1, 128, 192, 33, 31, 3, 1, 1), (2, 64, 128, 24, 24, 1, 1, 0),
                                                      (2, 64, 64, 32, 32, 3, 2, 1), (1, 32, 64, 20, 20, 3, 1, 1), (1, 64, 32, 64, 64, 3, 1, 1),
                                                      (8, 128, 128, 64, 64, 3, 1, 1), (1, 96, 48, 17, 19, 3, 1, 1),
                                                      # the tap-reuse kernel (3 x 3, stride 1, C % 64 == 0, W >= 16): rows that are not a multiple of the 16-pixel
                                                      # step (the x = 0 / x = W - 1 masks fall anywhere in a step), the narrowest map, one image row, a pixel count
                                                      # that is not a multiple of 16, 48 / 144 columns, several images
                                                      (3, 64, 48, 5, 21, 3, 1, 1), (2, 128, 144, 7, 16, 3, 1, 1), (1, 64, 64, 1, 37, 3, 1, 1), (5, 192, 128, 3, 17, 3, 1, 1),
                                                      (2, 64, 128, 19, 16, 3, 1, 1)])
def test_weight_gradient_with_pieces_of_both_operands(dev, FF, n, c, k, h, w, ksz, stride, pad):
    """fmi_conv2d_wgrad_f32 given the piece images of x (d.x3) and of dy (d.y3) against the same entry without them (2e-6 of the largest
    entry: same exact products, another summation order across pixel splits) and against torch fp32 autograd on the CPU (1e-5)"""
    lib = _lib()
    st = FF._st()
    g = torch.Generator().manual_seed(c + 5 * k + h)
    x = torch.randn(n, h, w, c, generator=g)
    wt_ = torch.randn(k, c, ksz, ksz, generator=g) / (c * ksz * ksz) ** 0.5
    xd = x.to(dev)
    d, oh, ow = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad)
    gy = torch.randn(n, oh, ow, k, generator=g)
    gyd = gy.to(dev)
    x3, _ = _split(FF, xd)
    gy3, _ = _split(FF, gyd)
    outs = []
    for use in (False, True):
        d, _, _ = FF.conv_desc(n, h, w, c, k, ksz, ksz, stride, pad)
        if use:
            d.x3, d.y3 = x3.data_ptr(), gy3.data_ptr()
        dw = torch.zeros(ksz * ksz, c, k, device=dev)
        lib.conv2d_wgrad_f32(C.byref(d), FF._p(xd), FF._p(gyd), FF._p(dw), None, 1, 0, st)
        outs.append(dw)
    scale = float(outs[0].abs().max())
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=2e-6 * scale)
    wr = wt_.clone().requires_grad_(True)
    F.conv2d(x.permute(0, 3, 1, 2), wr, None, stride=stride, padding=pad).backward(gy.permute(0, 3, 1, 2))
    ref = wr.grad.permute(2, 3, 1, 0).reshape(ksz * ksz, c, k)
    torch.testing.assert_close(outs[1].cpu(), ref, rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("n,h,w,c1,c2,cb", [(2, 128, 128, 32, 64, 32), (1, 130, 254, 64, 32, 32), (3, 96, 120, 32, 32, 64), (2, 128, 128, 16, 48, 28)])
def test_conv_transpose_pair_equals_two_calls(dev, FF, n, h, w, c1, c2, cb):
    """fmi_conv_transpose2d_pair_f32 (ResBlockDecoder's main path + bypass as one tap-reuse launch over all four sub-pixel phases, csrc/convt3x3.h)
    against the two ConvTranspose2d calls it replaces (second with the first's result as residual) -- result 2e-6 of the largest entry (same
    exact products, another summation order), gradients of both inputs, both weights and the bias 1e-5; and against torch on the CPU (1e-5).
    Shapes: ragged rows (the last pixel of an image row is cleared for the x + 1 view), 64 output columns (two column tiles), 28 outputs."""
    g = torch.Generator().manual_seed(n + h + c1)
    x1 = torch.randn(n, h, w, c1, generator=g).to(dev).requires_grad_(True)
    x2 = torch.randn(n, h, w, c2, generator=g).to(dev).requires_grad_(True)
    w1 = (torch.randn(c1, cb, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)   # nn.ConvTranspose2d weight: [in][out][kh][kw]
    w2 = (torch.randn(c2, cb, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = torch.randn(cb, generator=g).to(dev).requires_grad_(True)
    up = torch.randn(n, 2 * h, 2 * w, cb, generator=g).to(dev)
    res = {}
    for pair in (True, False):
        for t in (x1, x2, w1, w2, b):
            t.grad = None
        pw1, pw2 = FF.prepare_weights([(w1, None, None), (w2, None, None)])
        if pair:
            assert FF.conv_transpose2d_pair_ok(x1, pw1, x2, pw2)
            y = FF.conv_transpose2d_pair(x1, pw1, x2, pw2, b)
        else:
            y = FF.conv_transpose2d(x1, pw1, b, residual=FF.conv_transpose2d(x2, pw2))
        (y * up).sum().backward()
        res[pair] = [y.detach().clone()] + [t.grad.clone() for t in (x1, x2, w1, w2, b)]
    torch.testing.assert_close(res[True][0], res[False][0], rtol=0, atol=2e-6 * float(res[False][0].abs().max()))
    for a_, b_ in zip(res[True][1:], res[False][1:]):
        torch.testing.assert_close(a_, b_, rtol=0, atol=1e-5 * float(b_.abs().max()))
    ref = (F.conv_transpose2d(x1.detach().cpu().permute(0, 3, 1, 2), w1.detach().cpu(), None, stride=2, padding=1, output_padding=1) +
           F.conv_transpose2d(x2.detach().cpu().permute(0, 3, 1, 2), w2.detach().cpu(), b.detach().cpu(), stride=2, padding=1, output_padding=1))
    torch.testing.assert_close(res[True][0].cpu(), ref.permute(0, 2, 3, 1), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
