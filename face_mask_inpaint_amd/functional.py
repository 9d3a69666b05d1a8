"""Autograd glue between torch tensors and the HIP kernels of libfmi_hip.so.

Everything here works on *NHWC* fp32 device tensors (shape ``[N, H, W, C]``, contiguous).  torch is used for
device memory, the caching allocator, streams and the autograd tape only -- every floating-point operation of
the hot path is a call into the C ABI (``include/fmi_hip.h``).  There is no CPU path: a missing library or a
CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, FmiError

EW_LRELU, EW_LRELU_BWD, EW_TANH_BWD, EW_ADD, EW_SCALE, EW_AXPY, EW_MUL, EW_RELU_BWD_OUT, EW_SOFTPLUS, EW_SOFTPLUS_BWD, EW_SUB, EW_RSQRT, EW_RSQRT_BWD, EW_SIGMOID, EW_SIGMOID_BWD = range(15)
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2

# scratch budget for one attention score chunk (kept well inside the 256 MB Infinity Cache)
ATTN_CHUNK_BYTES = 96 * 1024 * 1024

# bench.py sets PROFILE = [] for one untimed step: every launch of the fp32-MFMA GEMM family is then bracketed by
# events on the launch stream and recorded as (call site, algorithmic flops, start, end).
PROFILE = None


class _prof:
    def __init__(self, tag, flops):
        self.tag, self.flops = tag, flops

    def __enter__(self):
        if PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e.record()
            PROFILE.append((self.tag, float(self.flops), self.s, self.e))
        return False


class _ZeroSlab:
    """Bump allocator of pre-zeroed device memory for the many small accumulators the kernels need (atomically accumulated weight /
    bias gradients, norm statistics, loss scalars): one 32 MB fill replaces hundreds of 5-microsecond fill launches per
    training step.  A slice is handed out once and never recycled, so it is zero when its kernel starts; the slab's storage lives as
    long as any slice does."""

    SLAB_BYTES = 32 << 20
    MAX_BYTES = 4 << 20
    buf = None
    off = 0

    @classmethod
    def take(cls, shape, device, dtype=torch.float32):
        if isinstance(shape, int):
            shape = (shape,)
        numel = 1
        for d in shape:
            numel *= int(d)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        # under HIP-graph capture every accumulator must be zeroed INSIDE the graph (a memset node per replay): a slab slice is zero
        # only the first time it is handed out
        if nbytes > cls.MAX_BYTES or nbytes == 0 or device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, device=device, dtype=dtype)
        nbytes_al = (nbytes + 255) & ~255
        if cls.buf is None or cls.buf.device != device or cls.off + nbytes_al > cls.SLAB_BYTES:
            cls.buf = torch.zeros(cls.SLAB_BYTES, device=device, dtype=torch.uint8)
            cls.off = 0
        out = cls.buf[cls.off:cls.off + nbytes].view(dtype).view(shape)
        cls.off += nbytes_al
        return out


def _zeros(shape, device, dtype=torch.float32):
    return _ZeroSlab.take(tuple(shape) if not isinstance(shape, int) else shape, device, dtype)


def _zeros_like(t):
    return _ZeroSlab.take(tuple(t.shape), t.device, t.dtype)


def _L():
    return _lib.lib()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor], off: int = 0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + 4 * off)


def _chk(*ts, dtype=torch.float32):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise FmiError("face_mask_inpaint_amd ops need device tensors (there is no CPU fallback)")
        if t.dtype != dtype:
            raise FmiError(f"expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise FmiError("expected a contiguous tensor")


class deterministic:
    """``with FF.deterministic():`` -- the library's reproducible mode (include/fmi_hip.h: fmi_set_deterministic; FMI_DETERMINISTIC=1 sets it
    for a whole process): single-contributor reductions instead of fp32 atomics from many workgroups, so two runs are bit-identical.
    Slower; a checking mode."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _L().set_deterministic(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        _L().set_deterministic(self.prev)
        return False


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped tensor -> NHWC contiguous (free when x is already channels-last in memory)."""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    """NHWC contiguous -> NCHW-shaped view (channels_last strides, no copy)."""
    return x.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------------
# raw wrappers (no autograd)
# ---------------------------------------------------------------------------------------------------
def gemm_raw(a_ptr, b_ptr, c_ptr, M, N, K, sa, sb, sc, batch=1, bs=(0, 0, 0), alpha=1.0, beta=0.0, bias=None, tag="gemm"):
    with _prof(f"{tag}|{M}x{N}x{K} b{batch}", 2.0 * M * N * K * batch):
        _L().gemm_f32(a_ptr, b_ptr, c_ptr, M, N, K, sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], batch, bs[0], bs[1], bs[2],
                      alpha, beta, _p(bias), _st())


def eltwise(op: int, a: torch.Tensor, b: Optional[torch.Tensor] = None, p0: float = 0.0, out: Optional[torch.Tensor] = None):
    _chk(a, b)
    y = torch.empty_like(a) if out is None else out
    if a.numel():
        _L().eltwise_f32(op, _p(a), _p(b), _p(y), a.numel(), float(p0), _st())
    return y


def conv_desc(n, h, w, c, k, kh, kw, stride, pad, pad_mode=0, x_cs=None, y_cs=None, dil=1, w3=None, x3=None) -> Tuple[ConvDesc, int, int]:
    """w3: the bf16 piece images (PackedWeight.wf3 / .wt3) of the weight operand the call will be given, or None; x3: the piece image
    (p3_of) of the ACTIVATION operand the call will be given (x for fwd, dy for dgrad), or None"""
    oh = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    d = ConvDesc(n, h, w, c, oh, ow, k, x_cs or c, y_cs or k, kh, kw, stride, pad, pad_mode, dil)
    if w3 is not None:
        d.w3 = w3.data_ptr()
        if x3 is not None:
            d.x3 = x3.data_ptr()
    return d, oh, ow


# ---------------------------------------------------------------------------------------------------
# bf16 piece images of activations (csrc/conv_p3.h): x3[pixel][C/16][3][16], x = x0 + x1 + x2 exactly.  With the pieces of BOTH operands
# at hand a convolution runs without split arithmetic in its inner loop.  An image is cut once per tensor (by the tensor's producer where
# that is a kernel of this library, by a fmi_split3_f32 pass otherwise) and cached on the tensor object: the forward / weight-gradient
# pair of x and the adjoint / weight-gradient pair of dy share it.
# ---------------------------------------------------------------------------------------------------
P3_ENABLED = True
P3_MIN_PIXELS = 16384  # smaller feature maps are launch / latency bound: the in-wave split costs nothing there
P3_MIN_WORK = int(os.environ.get("FMI_P3_MIN_WORK", "1152"))  # taps * output channels per activation element below which a split pass costs more than it returns
P3_MIN_COUT = int(os.environ.get("FMI_P3_MIN_COUT", "64"))    # (both overridable for A/B runs: tools/bench_tools/README.md)


def p3_wanted(pixels: int, cred: int, cout: int, taps: int) -> bool:
    """is the piece-image path worth one split pass over the activation operand?  (reduction channels cred, output channels cout)"""
    # one split pass moves 10 B per element (2.3 ps at 4.4 TB/s); the convolution spends 2 * taps * cout FLOP per element (3 x 3, 64 outputs:
    # 7.7 ps at 150 TFLOP/s) and gains ~20 % from the pieces: worth it from taps * cout = 1152 up (3 x 3 with 128 outputs)
    return P3_ENABLED and cred % 16 == 0 and cred >= 64 and cout >= P3_MIN_COUT and pixels >= P3_MIN_PIXELS and taps * cout >= P3_MIN_WORK


def p3_of(t: torch.Tensor, lrelu_from: Optional[torch.Tensor] = None, slope: float = 0.0, colsum: Optional[list] = None) -> Optional[torch.Tensor]:
    """piece image of the dense NHWC fp32 tensor t (cached on t).  lrelu_from: t is still EMPTY and becomes lrelu(lrelu_from, slope) in the
    same pass that cuts its pieces.  colsum: an empty list; when the pass over t (a gradient) actually runs here it also sums t over the
    pixels -- the bias gradient -- and the [C] tensor is appended (not in the reproducible mode: that sum uses atomics)."""
    c = t.shape[-1]
    if c % 16 or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.data_ptr() % 16:
        if lrelu_from is not None:
            eltwise(EW_LRELU, lrelu_from, None, slope, out=t)
        return None
    p = getattr(t, "_fmi_p3", None)
    if p is not None and p[1] == t._version and lrelu_from is None:
        return p[0]
    x3 = torch.empty(t.numel() * 3, device=t.device, dtype=torch.bfloat16)
    if lrelu_from is not None:
        _L().split3_f32(_p(lrelu_from), C.c_void_p(x3.data_ptr()), _p(t), t.numel() // c, c, 1, float(slope), _st())
    elif colsum is not None and c <= 512 and 256 % (c // 8) == 0 and not _L().get_deterministic():
        gb = _zeros(c, t.device, torch.float32)
        _L().split3_colsum_f32(_p(t), C.c_void_p(x3.data_ptr()), _p(gb), t.numel() // c, c, _st())
        colsum.append(gb)
    else:
        _L().split3_f32(_p(t), C.c_void_p(x3.data_ptr()), None, t.numel() // c, c, 0, 0.0, _st())
    t._fmi_p3 = (x3, t._version)
    return x3


# ---------------------------------------------------------------------------------------------------
# weight preparation (spectral norm + packing) for a whole network in one launch
# ---------------------------------------------------------------------------------------------------
class PackedWeight:
    """Per-call packed effective weight of one conv: wf [taps][C][K] (autograd tensor), wt [taps][K][C]."""

    __slots__ = ("wf", "wt", "rows", "C", "kh", "kw", "w3")

    def __init__(self, wf, wt, rows, Cc, kh, kw, w3=(None, None)):
        self.wf, self.wt, self.rows, self.C, self.kh, self.kw = wf, wt, rows, Cc, kh, kw
        self.w3 = w3  # (wf3, wt3): the same two packs as bf16 piece images (fmi_conv_desc.w3), or None where the shape has none


W3_ENABLED = True  # write the bf16 piece images of every eligible pack (False: the kernels split the weight fragments themselves)
_LAST_W3: List[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = []


class _WeightPrepare(torch.autograd.Function):
    """outputs: sigma[n], wf_0..wf_{n-1} (differentiable), wt_0..wt_{n-1} (twins for the adjoint kernels)."""

    @staticmethod
    def forward(ctx, metas, *ws):
        # metas[i] = (u, v[, want_w3[, power_iterations]]) parameters or (None, None); ws[i] = weight in torch layout [rows][C][kh][kw]
        lib = _L()
        n = len(ws)
        dev = ws[0].device
        sig = torch.ones(n, device=dev, dtype=torch.float32)
        entries = (_lib.WeightEntry * n)()
        wfs, wts, w3s = [], [], []
        for i, (w, meta) in enumerate(zip(ws, metas)):
            u, v = meta[0], meta[1]
            want_w3 = meta[2] if len(meta) > 2 else True
            iters = int(meta[3]) if len(meta) > 3 else 1
            _chk(w, u, v)
            rows, cc, kh, kw = w.shape
            if u is not None and rows > 4096:
                raise FmiError("spectral-norm weight with more than 4096 rows")
            wf = torch.empty((kh * kw, cc, rows), device=dev, dtype=torch.float32)
            wt = torch.empty((kh * kw, rows, cc), device=dev, dtype=torch.float32)
            pieces = W3_ENABLED and want_w3 and kh * kw <= 36 and w.is_cuda
            wf3 = torch.empty(3 * kh * kw * cc * rows, device=dev, dtype=torch.bfloat16) if pieces and cc % 16 == 0 else None
            wt3 = torch.empty(3 * kh * kw * cc * rows, device=dev, dtype=torch.bfloat16) if pieces and rows % 16 == 0 else None
            w3s.append((wf3, wt3))
            e = entries[i]
            e.wf3 = wf3.data_ptr() if wf3 is not None else None
            e.wt3 = wt3.data_ptr() if wt3 is not None else None
            e.w = w.data_ptr()
            e.u = u.data_ptr() if u is not None else None
            e.v = v.data_ptr() if v is not None else None
            e.wf, e.wt = wf.data_ptr(), wt.data_ptr()
            e.sigma = sig.data_ptr() + 4 * i
            e.rows, e.C, e.taps, e.iters = rows, cc, kh * kw, iters
            wfs.append(wf)
            wts.append(wt)
        lib.weight_prepare_f32(entries, n, _st())
        ctx.metas, ctx.ws, ctx.sig = metas, ws, sig
        _LAST_W3[:] = w3s  # handed to prepare_weights (plain tensors next to the autograd outputs)
        ctx.mark_non_differentiable(sig, *wts)
        ctx.set_materialize_grads(False)  # packs that received no gradient arrive as None, not as freshly zero-filled tensors
        return (sig,) + tuple(wfs) + tuple(wts)

    @staticmethod
    def backward(ctx, _gsig, *gs):
        lib = _L()
        n = len(ctx.ws)
        gwf = gs[:n]
        live = [i for i, g in enumerate(gwf) if g is not None and ctx.needs_input_grad[1 + i]]
        grads: List[Optional[torch.Tensor]] = [None] * n
        if live:
            entries = (_lib.WeightGradEntry * len(live))()
            keep = []
            for j, i in enumerate(live):
                w = ctx.ws[i]
                u, v = ctx.metas[i][0], ctx.metas[i][1]
                g = gwf[i].contiguous()
                dw = torch.empty_like(w)
                keep.append(g)
                e = entries[j]
                e.w = w.data_ptr()
                # u / v are read LIVE (not snapshotted): this reproduces the reference's `.data` rebinding,
                # see DESIGN.md "SpectralNorm state" and oracle/picnet_cpu.py:sn_weight
                e.u = u.data_ptr() if u is not None else None
                e.v = v.data_ptr() if v is not None else None
                e.sigma = ctx.sig.data_ptr() + 4 * i
                e.dwf = g.data_ptr()
                e.dw = dw.data_ptr()
                e.rows, e.C, e.taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                grads[i] = dw
            scratch = _zeros(len(live), ctx.sig.device, torch.float32)
            lib.weight_grad_f32(entries, len(live), _p(scratch), _st())
        return (None,) + tuple(grads)


def prepare_weights(items: Sequence[tuple]) -> List[PackedWeight]:
    """items: (weight[rows,C,kh,kw], u or None, v or None[, want_w3 = True[, power_iterations = 1]]).  One launch sequence; returns
    PackedWeight per item.  want_w3 = False: no bf16 piece images for this weight (its convolution runs on bf16 activations, which has
    its own weight packs)."""
    metas = tuple(tuple(it[1:]) for it in items)
    ws = tuple(it[0] for it in items)
    n = len(ws)
    res = _WeightPrepare.apply(metas, *ws)
    w3s = list(_LAST_W3)
    _LAST_W3.clear()
    return [PackedWeight(res[1 + i], res[1 + n + i], w.shape[0], w.shape[1], w.shape[2], w.shape[3], w3s[i]) for i, w in enumerate(ws)]


# ---------------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------------
def _act_bwd(g, y, act):
    if act == ACT_TANH:
        return eltwise(EW_TANH_BWD, g, y)
    if act == ACT_RELU:
        return eltwise(EW_RELU_BWD_OUT, g, y)
    return g


class _Conv2d(torch.autograd.Function):
    """y = act(conv(x, W) + bias + residual); x [N,H,W,C], wf [taps][C][K]."""

    @staticmethod
    def forward(ctx, x, wf, bias, residual, wt, kh, kw, stride, pad, pad_mode, act, in_act=None, skip_act_bwd=False, dil=1, passthrough=False, w3=(None, None)):
        """in_act: None, ("apply", slope): the convolution reads lrelu(x, slope) (computed here, and only IT is kept for the backward),
        ("mask", slope): x already is the output of such an activation; either way the input gradient is multiplied by act'(x) in the
        adjoint's epilogue.  skip_act_bwd: this convolution's own fused activation (act) is differentiated by its single consumer (a
        following convolution with in_act = ("mask", .)), not here."""
        _chk(x, wf, bias, residual)
        lib = _L()
        x_in = x
        n, h, w, c = x.shape
        k = wf.shape[2]
        use3 = w3[0] is not None and pad_mode == 0 and p3_wanted(n * h * w, c, k, kh * kw)
        x3 = None
        if in_act is not None and in_act[0] == "apply":
            if use3:  # lrelu(x) and its piece image in one pass
                x = torch.empty_like(x_in)
                x3 = p3_of(x, lrelu_from=x_in, slope=in_act[1])
            else:
                x = eltwise(EW_LRELU, x, None, in_act[1])
        elif use3:
            x3 = p3_of(x)
        d, oh, ow = conv_desc(n, h, w, c, k, kh, kw, stride, pad, pad_mode, dil=dil, w3=w3[0], x3=x3)
        ctx.wt3 = w3[1]
        ctx.x3 = x3
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=torch.float32)
        with _prof(f"conv_fwd|{n}x{h}x{w} {c}->{k} k{kh}s{stride}" + (f"d{dil}" if dil > 1 else ""), 2.0 * n * oh * ow * k * c * kh * kw):
            lib.conv2d_fwd_f32(C.byref(d), _p(x), _p(wf), _p(bias), _p(residual), _p(y), act, 1, 0, _st())
        ctx.save_for_backward(x, wf, y if (act and not skip_act_bwd) else None)
        ctx.wt, ctx.cfg, ctx.has = wt, (kh, kw, stride, pad, pad_mode, act), (bias is not None, residual is not None)
        ctx.dil = dil
        ctx.in_slope = None if in_act is None else float(in_act[1])
        ctx.skip_act_bwd = bool(skip_act_bwd)
        if passthrough:  # the input handed on to its OTHER consumer: that consumer's gradient comes back into this backward
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, gy, gpass=None):
        lib = _L()
        x, wf, y = ctx.saved_tensors
        if gy is None:
            return (gpass,) + (None,) * 15
        kh, kw, stride, pad, pad_mode, act = ctx.cfg
        gy = gy.contiguous()
        if act and not ctx.skip_act_bwd:
            gy = _act_bwd(gy, y, act)
        n, h, w, c = x.shape
        k = wf.shape[2]
        gx = gwf = gb = gres = None
        masked = False
        dil = ctx.dil
        gb_of_split = [] if (ctx.has[0] and ctx.needs_input_grad[2]) else None  # filled by the pass that cuts dy's pieces, if one runs
        if ctx.needs_input_grad[0]:
            d0, _, _ = conv_desc(n, h, w, c, k, kh, kw, stride, pad, pad_mode, dil=dil)
            if pad_mode == 1 and lib.conv2d_thin_supported(C.byref(d0)):  # thin output: adjoint and fold in one pass
                gx = torch.empty_like(x)
                with _prof(f"conv_dgrad|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                    lib.conv2d_thin_dgrad_f32(C.byref(d0), _p(gy), _p(ctx.wt), _p(gx), _st())
            elif pad_mode == 1:  # adjoint w.r.t. the reflection-padded tensor, then fold (base_function.py:390)
                hp, wp = h + 2 * pad, w + 2 * pad
                d, _, _ = conv_desc(n, hp, wp, c, k, kh, kw, stride, 0, w3=ctx.wt3)
                gpad = torch.empty((n, hp, wp, c), device=x.device, dtype=torch.float32)
                with _prof(f"conv_dgrad|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                    lib.conv2d_dgrad_f32(C.byref(d), _p(gy), _p(ctx.wt), None, None, _p(gpad), 1, 0, _st())
                gx = torch.empty_like(x)
                lib.reflect_pad_fold_f32(_p(gpad), _p(gx), n, h, w, c, pad, _st())
            else:
                oh_, ow_ = gy.shape[1], gy.shape[2]
                gy3 = p3_of(gy, colsum=gb_of_split) if (ctx.wt3 is not None and p3_wanted(n * oh_ * ow_, k, c, kh * kw)) else None
                d, _, _ = conv_desc(n, h, w, c, k, kh, kw, stride, pad, dil=dil, w3=ctx.wt3, x3=gy3)
                gx = torch.empty_like(x)
                with _prof(f"conv_dgrad|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                    if ctx.in_slope is not None and gpass is not None:  # act'(x) in the epilogue, then + the other consumer's gradient
                        lib.conv2d_dgrad_masked_add_f32(C.byref(d), _p(gy), _p(ctx.wt), _p(x), ctx.in_slope, _p(gpass.contiguous()), _p(gx), _st())
                        masked, gpass = True, None
                    elif ctx.in_slope is not None:  # act'(x) folded into the adjoint's epilogue
                        lib.conv2d_dgrad_masked_f32(C.byref(d), _p(gy), _p(ctx.wt), _p(x), ctx.in_slope, _p(gx), _st())
                        masked = True
                    elif gpass is not None:
                        lib.conv2d_dgrad_f32(C.byref(d), _p(gy), _p(ctx.wt), None, _p(gpass.contiguous()), _p(gx), 1, 0, _st())
                        gpass = None
                    else:
                        lib.conv2d_dgrad_f32(C.byref(d), _p(gy), _p(ctx.wt), None, None, _p(gx), 1, 0, _st())
            if ctx.in_slope is not None and not masked:
                gx = eltwise(EW_LRELU_BWD, gx, x, ctx.in_slope)
            if gpass is not None:  # paths without a fused epilogue (thin / reflect)
                gx = eltwise(EW_ADD, gx, gpass.contiguous())
        elif gpass is not None:
            gx = gpass
        if ctx.needs_input_grad[1]:
            d, _, _ = conv_desc(n, h, w, c, k, kh, kw, stride, pad, pad_mode, dil=dil)
            gwf = _zeros_like(wf)
            want_gb = ctx.has[0] and ctx.needs_input_grad[2]
            # both operands as piece images (x's from the forward, dy's shared with the adjoint above): the weight gradient runs without
            # split arithmetic; the bias gradient then takes its own pass
            if ctx.x3 is not None and pad_mode == 0 and c % 32 == 0 and k % 16 == 0 and p3_wanted(gy.numel() // k, c, k, kh * kw):
                gy3 = p3_of(gy, colsum=gb_of_split)
                if gy3 is not None:
                    d.w3 = None
                    d.x3, d.y3 = ctx.x3.data_ptr(), gy3.data_ptr()
                    want_gb_fused = False
                else:
                    want_gb_fused = want_gb
            else:
                want_gb_fused = want_gb
            fuse = want_gb_fused and (kh * kw * c) % 4 == 0
            if fuse:
                gb = _zeros(k, x.device, torch.float32)
            with _prof(f"conv_wgrad|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                lib.conv2d_wgrad_f32(C.byref(d), _p(x), _p(gy), _p(gwf), _p(gb) if fuse else None, 1, 0, _st())
        if ctx.has[0] and ctx.needs_input_grad[2] and gb is None:
            if gb_of_split:  # summed by the pass that cut dy's pieces
                gb = gb_of_split[0]
            else:
                gb = _zeros(k, x.device, torch.float32)
                lib.bias_grad_f32(_p(gy), gy.numel() // k, k, k, _p(gb), _st())
        if ctx.has[1] and ctx.needs_input_grad[3]:
            gres = gy
        return gx, gwf, gb, gres, None, None, None, None, None, None, None, None, None, None, None, None


class _ThinConvLReLU(torch.autograd.Function):
    """y = act(conv3x3(lrelu(x, slope), W) + bias) for a thin-output (K <= 4) stride-1 pad-1 convolution of a 32-channel map: the
    generator's Output block (base_function.py:386-396) with the LeakyReLU folded into the convolution kernels -- x is read once per
    direction, lrelu(x) is never written."""

    @staticmethod
    def forward(ctx, x, wf, bias, wt, slope, pad_mode, act):
        _chk(x, wf, bias)
        n, h, w, c = x.shape
        k = wf.shape[2]
        d, oh, ow = conv_desc(n, h, w, c, k, 3, 3, 1, 1, pad_mode)
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=torch.float32)
        with _prof(f"conv_fwd|{n}x{h}x{w} {c}->{k} k3s1", 2.0 * n * oh * ow * k * c * 9):
            _L().conv2d_thin_lrelu_fwd_f32(C.byref(d), _p(x), slope, _p(wf), _p(bias), _p(y), act, _st())
        ctx.save_for_backward(x, wf, y if act else None)
        ctx.wt, ctx.cfg, ctx.has_b = wt, (slope, pad_mode, act), bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _L()
        x, wf, y = ctx.saved_tensors
        slope, pad_mode, act = ctx.cfg
        gy = gy.contiguous()
        if act:
            gy = _act_bwd(gy, y, act)
        n, h, w, c = x.shape
        k = wf.shape[2]
        d, _, _ = conv_desc(n, h, w, c, k, 3, 3, 1, 1, pad_mode)
        gx = gwf = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            with _prof(f"conv_dgrad|{n}x{h}x{w} {c}->{k} k3s1", 2.0 * gy.numel() * c * 9):
                lib.conv2d_thin_lrelu_dgrad_f32(C.byref(d), _p(gy), _p(ctx.wt), _p(x), slope, _p(gx), _st())
        if ctx.needs_input_grad[1]:
            gwf = _zeros_like(wf)
            if ctx.has_b and ctx.needs_input_grad[2]:
                gb = _zeros(k, x.device, torch.float32)
            with _prof(f"conv_wgrad|{n}x{h}x{w} {c}->{k} k3s1", 2.0 * gy.numel() * c * 9):
                lib.conv2d_thin_lrelu_wgrad_f32(C.byref(d), _p(x), slope, _p(gy), _p(gwf), _p(gb), _st())
        elif ctx.has_b and ctx.needs_input_grad[2]:
            gb = _zeros(k, x.device, torch.float32)
            lib.bias_grad_f32(_p(gy), gy.numel() // k, k, k, _p(gb), _st())
        return gx, gwf, gb, None, None, None, None


def lrelu_conv2d(x, pw: PackedWeight, bias=None, slope=0.1, pad=1, pad_mode=0, act=ACT_NONE):
    """act(conv(lrelu(x, slope))): one fused pass when the thin-output kernels take the shape, LeakyReLU + conv2d otherwise"""
    if x.dtype == torch.float32 and pw.kh == 3 and pw.kw == 3 and pad == 1 and x.is_cuda:
        n, h, w, c = x.shape
        d, _, _ = conv_desc(n, h, w, c, pw.wf.shape[2], 3, 3, 1, 1, pad_mode)
        if _L().conv2d_thin_lrelu_supported(C.byref(d)) and x.data_ptr() % 16 == 0:
            return _ThinConvLReLU.apply(x, pw.wf, bias, pw.wt, float(slope), pad_mode, act)
    return conv2d(leaky_relu(x, slope), pw, bias, None, 1, pad, pad_mode, act)


def conv2d(x, pw: PackedWeight, bias=None, residual=None, stride=1, pad=0, pad_mode=0, act=ACT_NONE, in_act=None, skip_act_bwd=False, dilation=1,
           passthrough=False):
    """passthrough=True: returns (y, x') with x' = x for x's other consumer (see _Conv2d)"""
    if x.dtype == BF16:
        if bias is not None or residual is not None or pad_mode or act or in_act is not None or dilation != 1:
            raise FmiError("the bf16 convolution has no bias / residual / activation / reflect-padding / dilation form")
        return _Conv2dBF16.apply(x, pw.wf, pw.wt, pw.kh, pw.kw, stride, pad)
    return _Conv2d.apply(x, pw.wf, bias, residual, pw.wt, pw.kh, pw.kw, stride, pad, pad_mode, act, in_act, skip_act_bwd, int(dilation), bool(passthrough), pw.w3)


# ---- bf16 activations (StyleGAN2 decoder of configs C3 / C5): fp32 master weights, bf16 copies packed per call ----
BF16 = torch.bfloat16


def _split_ws(out: torch.Tensor, kred: int):
    """zeroed fp32 twin of a SMALL bf16 convolution output with a deep reduction (the 4^2 .. 16^2 decoder layers): lets the library
    split the reduction over workgroups; None (= never split) otherwise"""
    if out.numel() > (1 << 22) or kred < 1024:
        return None, 0
    ws = torch.zeros(out.numel(), device=out.device, dtype=torch.float32)
    return ws, ws.numel()


def _pack_bf16(src_tab: torch.Tensor) -> torch.Tensor:
    """fp32 [T][A][B] -> bf16 [B][T][A] (reduction index contiguous): wf -> [K][taps][C], wt -> [C][taps][K]"""
    t, a, b = src_tab.shape
    out = torch.empty((b, t, a), device=src_tab.device, dtype=BF16)
    _L().pack_weight_bf16(_p(src_tab), _p(out), t, a, b, _st())
    return out


class _Conv2dBF16(torch.autograd.Function):
    """y = conv(x, W) on bf16 NHWC activations, fp32 accumulation; wf / wt are the fp32 packs (wf carries the weight gradient)."""

    @staticmethod
    def forward(ctx, x, wf, wt, kh, kw, stride, pad):
        _chk(x, dtype=BF16)
        _chk(wf, wt)
        lib = _L()
        n, h, w, c = x.shape
        k = wf.shape[2]
        d, oh, ow = conv_desc(n, h, w, c, k, kh, kw, stride, pad)
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=BF16)
        wnk = _pack_bf16(wf)
        ws, wsn = _split_ws(y, c * kh * kw)
        with _prof(f"conv_fwd_bf16|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * n * oh * ow * k * c * kh * kw):
            lib.conv2d_fwd_bf16(C.byref(d), _p(x), _p(wnk), None, _p(y), _p(ws), wsn, _st())
        ctx.save_for_backward(x, wf, wt)
        ctx.cfg = (kh, kw, stride, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _L()
        x, wf, wt = ctx.saved_tensors
        kh, kw, stride, pad = ctx.cfg
        gy = gy.contiguous()
        n, h, w, c = x.shape
        k = wf.shape[2]
        d, _, _ = conv_desc(n, h, w, c, k, kh, kw, stride, pad)
        gx = gwf = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wck = _pack_bf16(wt)
            ws, wsn = _split_ws(gx, k * kh * kw)
            with _prof(f"conv_dgrad_bf16|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                lib.conv2d_dgrad_bf16(C.byref(d), _p(gy), _p(wck), None, _p(gx), _p(ws), wsn, _st())
        if ctx.needs_input_grad[1]:
            gwf = _zeros_like(wf)
            with _prof(f"conv_wgrad_bf16|{n}x{h}x{w} {c}->{k} k{kh}s{stride}", 2.0 * gy.numel() * c * kh * kw):
                lib.conv2d_wgrad_bf16(C.byref(d), _p(x), _p(gy), _p(gwf), _st())
        return gx, gwf, None, None, None, None, None


class _ConvTranspose2dBF16(torch.autograd.Function):
    """bf16 twin of _ConvTranspose2d: the adjoint of the stride-s conv view, sub-pixel phases inside the library."""

    @staticmethod
    def forward(ctx, x, wf, wt, kh, kw, stride, pad, out_pad):
        _chk(x, dtype=BF16)
        _chk(wf, wt)
        lib = _L()
        n, h, w, cs = x.shape
        cb = wf.shape[1]
        H = (h - 1) * stride - 2 * pad + kh + out_pad
        W = (w - 1) * stride - 2 * pad + kw + out_pad
        d, oh, ow = conv_desc(n, H, W, cb, cs, kh, kw, stride, pad)
        if (oh, ow) != (h, w):
            raise FmiError("unsupported ConvTranspose2d geometry")
        y = torch.empty((n, H, W, cb), device=x.device, dtype=BF16)
        wck = _pack_bf16(wt)
        ws, wsn = _split_ws(y, cs * kh * kw // (stride * stride))
        with _prof(f"convT_fwd_bf16|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
            lib.conv2d_dgrad_bf16(C.byref(d), _p(x), _p(wck), None, _p(y), _p(ws), wsn, _st())
        ctx.save_for_backward(x, wf)
        ctx.cfg, ctx.HW = (kh, kw, stride, pad), (H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _L()
        x, wf = ctx.saved_tensors
        kh, kw, stride, pad = ctx.cfg
        H, W = ctx.HW
        gy = gy.contiguous()
        n, h, w, cs = x.shape
        cb = wf.shape[1]
        d, _, _ = conv_desc(n, H, W, cb, cs, kh, kw, stride, pad)
        gx = gwf = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wnk = _pack_bf16(wf)
            ws, wsn = _split_ws(gx, cb * kh * kw)
            with _prof(f"convT_dgrad_bf16|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
                lib.conv2d_fwd_bf16(C.byref(d), _p(gy), _p(wnk), None, _p(gx), _p(ws), wsn, _st())
        if ctx.needs_input_grad[1]:
            gwf = _zeros_like(wf)
            with _prof(f"convT_wgrad_bf16|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
                lib.conv2d_wgrad_bf16(C.byref(d), _p(gy), _p(x), _p(gwf), _st())
        return gx, gwf, None, None, None, None, None, None


def _styled_out_bwd(g, y, noise, d, nw, bias, slope, gain, need_d, need_b, need_nw):
    """adjoint of the fused StyledConv output stage (fmi_styled_out_bwd_bf16): one pass over g and y -> (t, gd, gbias, gnw)"""
    n, c = y.shape[0], y.shape[-1]
    p = y.numel() // (n * c)
    t = torch.empty_like(y)
    gd = torch.empty((n, c), device=y.device, dtype=torch.float32) if (need_d and d is not None) else None
    gb = torch.empty(c, device=y.device, dtype=torch.float32) if (need_b and bias is not None) else None
    gnw = torch.empty(1, device=y.device, dtype=torch.float32) if (need_nw and noise is not None) else None
    ws = _parts_ws(y.device, max(2048, n) * 3 * c)
    sums = torch.empty(n * 3 * c, device=y.device, dtype=torch.float32)
    with _prof(f"bytes:noise_bias_act_fused_bwd|{tuple(y.shape)}", 3.0 * y.element_size() * y.numel()):
        _L().styled_out_bwd_bf16(_p(g), _p(y), _p(noise), _p(d), _p(nw), _p(bias), _p(t), _p(gd), _p(gb), _p(gnw), _p(ws), ws.numel(), _p(sums),
                                 n, p, c, slope, gain, _st())
    return t, gd, gb, gnw


class _StyledConvBF16(torch.autograd.Function):
    """y = lrelu(conv(x, W) * d[n][k] + nw * noise + bias, slope) * gain on bf16 NHWC activations in ONE forward launch (the output
    stage of the eight-phase kernel) and one fused adjoint pass in front of the convolution's own gradients: ModulatedConv2d on
    pre-scaled activations + demodulation + NoiseInjection + FusedLeakyReLU (stylegan2/model.py:241-294, op/fused_act.py:30-69)."""

    @staticmethod
    def forward(ctx, x, wf, wt, d, noise, nw, bias, kh, kw, pad, slope, gain):
        _chk(x, dtype=BF16)
        _chk(wf, wt, d, noise, nw, bias)
        lib = _L()
        n, h, w, c = x.shape
        k = wf.shape[2]
        dsc, oh, ow = conv_desc(n, h, w, c, k, kh, kw, 1, pad)
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=BF16)
        wnk = _pack_bf16(wf)
        with _prof(f"conv_fwd_bf16|{n}x{h}x{w} {c}->{k} k{kh}s1 +out", 2.0 * n * oh * ow * k * c * kh * kw):
            lib.conv2d_fwd_act_bf16(C.byref(dsc), _p(x), _p(wnk), _p(d), _p(noise), _p(nw), _p(bias), slope, gain, _p(y), _st())
        ctx.save_for_backward(x, wf, wt, d, noise, nw, bias, y)
        ctx.cfg = (kh, kw, pad, slope, gain)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _L()
        x, wf, wt, d, noise, nw, bias, y = ctx.saved_tensors
        kh, kw, pad, slope, gain = ctx.cfg
        n, h, w, c = x.shape
        k = wf.shape[2]
        ni = ctx.needs_input_grad
        t, gd, gb, gnw = _styled_out_bwd(g.contiguous(), y, noise, d, nw, bias, slope, gain, ni[3], ni[6], ni[5])
        dsc, _, _ = conv_desc(n, h, w, c, k, kh, kw, 1, pad)
        gx = gwf = None
        if ni[0]:
            gx = torch.empty_like(x)
            wck = _pack_bf16(wt)
            ws, wsn = _split_ws(gx, k * kh * kw)
            with _prof(f"conv_dgrad_bf16|{n}x{h}x{w} {c}->{k} k{kh}s1", 2.0 * t.numel() * c * kh * kw):
                lib.conv2d_dgrad_bf16(C.byref(dsc), _p(t), _p(wck), None, _p(gx), _p(ws), wsn, _st())
        if ni[1]:
            gwf = _zeros_like(wf)
            with _prof(f"conv_wgrad_bf16|{n}x{h}x{w} {c}->{k} k{kh}s1", 2.0 * t.numel() * c * kh * kw):
                lib.conv2d_wgrad_bf16(C.byref(dsc), _p(x), _p(t), _p(gwf), _st())
        return gx, gwf, None, gd, None, gnw, gb, None, None, None, None, None


def styled_conv_fused_ok(x, pw: PackedWeight, pad) -> bool:
    """shapes the fused StyledConv launch takes (the eight-phase kernel's conditions) and where it pays: the decoder's large maps"""
    if x.dtype != BF16 or not x.is_cuda:
        return False
    n, h, w, c = x.shape
    k = pw.wf.shape[2]
    oh, ow = h + 2 * pad - pw.kh + 1, w + 2 * pad - pw.kw + 1
    return c % 64 == 0 and k % 4 == 0 and k > 64 and pw.kh * pw.kw <= 32 and n * oh * ow >= 32768


def styled_conv(x, pw: PackedWeight, d, noise, nw, bias, slope, gain, pad):
    return _StyledConvBF16.apply(x, pw.wf, pw.wt, d, noise, nw, bias, pw.kh, pw.kw, int(pad), float(slope), float(gain))


class _BlurActBF16(torch.autograd.Function):
    """y = lrelu(Blur(u) * d[n][c] + nw * noise + bias, slope) * gain in one pass (fmi_blur_act_bf16): the Blur after an upsampling
    ModulatedConv2d with the demodulation, NoiseInjection and FusedLeakyReLU behind it (stylegan2/model.py:88-91, 250-252, 268-294)."""

    @staticmethod
    def forward(ctx, u, kernel, d, noise, nw, bias, pad0, pad1, slope, gain):
        _chk(u, dtype=BF16)
        _chk(kernel, d, noise, nw, bias)
        n, h, w, c = u.shape
        oh, ow = h + pad0 + pad1 - 3, w + pad0 + pad1 - 3
        y = torch.empty((n, oh, ow, c), device=u.device, dtype=BF16)
        with _prof(f"bytes:upfirdn2d|{n}x{h}x{w}x{c} up1 down1 +out", float(u.element_size()) * (u.numel() + y.numel())):
            _L().blur_act_bf16(_p(u), _p(kernel), _p(y), n, h, w, c, pad0, pad1, pad0, pad1, _p(d), _p(noise), _p(nw), _p(bias), slope, gain, 1, _st())
        ctx.save_for_backward(kernel, d, noise, nw, bias, y)
        ctx.cfg = (pad0, pad1, slope, gain, (n, h, w, c))
        return y

    @staticmethod
    def backward(ctx, g):
        kernel, d, noise, nw, bias, y = ctx.saved_tensors
        pad0, pad1, slope, gain, (n, h, w, c) = ctx.cfg
        ni = ctx.needs_input_grad
        t, gd, gb, gnw = _styled_out_bwd(g.contiguous(), y, noise, d, nw, bias, slope, gain, ni[2], ni[5], ni[4])
        gu = None
        if ni[0]:
            oh, ow = y.shape[1], y.shape[2]
            gk = torch.flip(kernel, [0, 1]).contiguous()
            g0, g1 = 3 - pad0, h - oh + pad0  # the adjoint FIR's pads (op/upfirdn2d.py:108-113 with up = down = 1)
            gu = torch.empty((n, h, w, c), device=g.device, dtype=BF16)
            with _prof(f"bytes:upfirdn2d_bwd|{n}x{h}x{w}x{c} up1 down1", float(t.element_size()) * (t.numel() + gu.numel())):
                _L().blur_act_bf16(_p(t), _p(gk), _p(gu), n, oh, ow, c, g0, g1, g0, g1, None, None, None, None, 1.0, 1.0, 1, _st())
        return gu, None, gd, None, gnw, gb, None, None, None, None


def blur_act(u, kernel, pad, d, noise, nw, bias, slope, gain):
    """kernel: the Blur's 4 x 4 taps (an outer product by construction)"""
    return _BlurActBF16.apply(u, kernel.contiguous(), d, noise, nw, bias, int(pad[0]), int(pad[1]), float(slope), float(gain))


def _parts_ws(device, floats: int) -> torch.Tensor:
    """partials workspace of the bf16 reduction passes (contents irrelevant, fully overwritten before it is read)"""
    return torch.empty(int(floats), device=device, dtype=torch.float32)


class _ToRGB(torch.autograd.Function):
    """out[n,p,o] = sum_c x[n,p,c] w[o,c] s[n,c] + bias[o] + skip[n,p,o]: the 1x1 modulated convolution of ToRGB without
    demodulation (stylegan2/model.py:349-369) on bf16 activations; w [3,C], s [N,C], bias [3], skip / out [N,H,W,3] are fp32."""

    @staticmethod
    def forward(ctx, x, w, s, bias, skip):
        _chk(x, dtype=BF16)
        _chk(w, s, bias, skip)
        n, h, wd, c = x.shape
        out = torch.empty((n, h, wd, 3), device=x.device, dtype=torch.float32)
        _L().torgb_fwd_bf16(_p(x), _p(w), _p(s), _p(bias), _p(skip), _p(out), n, h * wd, c, _st())
        ctx.save_for_backward(x, w, s)
        ctx.has = (bias is not None, skip is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, s = ctx.saved_tensors
        g = g.contiguous()
        n, h, wd, c = x.shape
        gx = torch.empty_like(x)
        ws = _parts_ws(x.device, (2048 + n) * (3 * c + 8))
        gw, gs = torch.empty_like(w), torch.empty_like(s)
        gb = torch.empty(3, device=x.device, dtype=torch.float32) if ctx.has[0] else None
        _L().torgb_bwd_bf16(_p(x), _p(w), _p(s), _p(g), _p(gx), _p(ws), ws.numel(), _p(gw), _p(gs), _p(gb), n, h * wd, c, _st())
        return gx, gw, gs, gb, (g if ctx.has[1] else None)


def torgb(x, w, s, bias=None, skip=None):
    return _ToRGB.apply(x, w.contiguous(), s.contiguous(), bias, skip)


_BIAS_GRAD_MEMO = [None, None, -1]  # (cotangent tensor, its column sums)


def _bias_grad_of(gy, cb):
    """sum of gy over all pixels.  The main and the bypass ConvTranspose2d of a ResBlockDecoder (base_function.py:308-364) are
    summed, so both receive the SAME cotangent tensor: the second request is answered from the first (one pass over up to
    1 GB instead of two).  The memo keeps that tensor alive, so its address cannot be recycled while it is the key."""
    m = _BIAS_GRAD_MEMO
    if m[0] is not None and m[0].data_ptr() == gy.data_ptr() and m[0].shape == gy.shape and m[0]._version == gy._version and m[2] == gy._version:
        return m[1].clone()
    gb = _zeros(cb, gy.device, torch.float32)
    _L().bias_grad_f32(_p(gy), gy.numel() // cb, cb, cb, _p(gb), _st())
    m[:] = [gy, gb, gy._version]
    return gb.clone()


class _ConvTranspose2d(torch.autograd.Function):
    """ConvTranspose2d as the adjoint of the stride-s conv that maps the big image onto the small one.
    x [N,h,w,Cs]; packed weights are those of the conv view (rows = Cs, C = Cb)."""

    @staticmethod
    def forward(ctx, x, wf, bias, residual, wt, kh, kw, stride, pad, out_pad, w3=(None, None)):
        _chk(x, wf, bias, residual)
        lib = _L()
        n, h, w, cs = x.shape
        cb = wf.shape[1]
        H = (h - 1) * stride - 2 * pad + kh + out_pad
        W = (w - 1) * stride - 2 * pad + kw + out_pad
        x3 = p3_of(x) if (w3[1] is not None and p3_wanted(n * h * w, cs, cb, kh * kw)) else None
        d, oh, ow = conv_desc(n, H, W, cb, cs, kh, kw, stride, pad, w3=w3[1], x3=x3)
        ctx.wf3 = w3[0]
        ctx.x3 = x3
        if (oh, ow) != (h, w):
            raise FmiError("unsupported ConvTranspose2d geometry")
        y = torch.empty((n, H, W, cb), device=x.device, dtype=torch.float32)
        with _prof(f"convT_fwd|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
            lib.conv2d_dgrad_f32(C.byref(d), _p(x), _p(wt), _p(bias), _p(residual), _p(y), 1, 0, _st())
        ctx.save_for_backward(x, wf)
        ctx.cfg, ctx.has, ctx.HW = (kh, kw, stride, pad), (bias is not None, residual is not None), (H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _L()
        x, wf = ctx.saved_tensors
        kh, kw, stride, pad = ctx.cfg
        H, W = ctx.HW
        gy = gy.contiguous()
        n, h, w, cs = x.shape
        cb = wf.shape[1]
        d, _, _ = conv_desc(n, H, W, cb, cs, kh, kw, stride, pad, w3=ctx.wf3)
        gx = gwf = gb = gres = None
        gb_of_split = [] if (ctx.has[0] and ctx.needs_input_grad[2]) else None  # filled by the pass that cuts dy's pieces, if one runs
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            gy3 = p3_of(gy, colsum=gb_of_split) if (ctx.wf3 is not None and p3_wanted(n * H * W, cb, cs, kh * kw)) else None
            d0, _, _ = conv_desc(n, H, W, cb, cs, kh, kw, stride, pad, w3=ctx.wf3, x3=gy3)
            with _prof(f"convT_dgrad|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
                lib.conv2d_fwd_f32(C.byref(d0), _p(gy), _p(wf), None, None, _p(gx), 0, 1, 0, _st())
        if ctx.needs_input_grad[1]:
            gwf = _zeros_like(wf)
            if ctx.x3 is not None and cb % 32 == 0 and cs % 16 == 0:  # the big image's pieces (shared with the adjoint above) and x's from the forward
                gy3 = p3_of(gy, colsum=gb_of_split)
                if gy3 is not None:
                    d.x3, d.y3 = gy3.data_ptr(), ctx.x3.data_ptr()
            with _prof(f"convT_wgrad|{n}x{h}x{w} {cs}->{cb}", 2.0 * x.numel() * cb * kh * kw):
                lib.conv2d_wgrad_f32(C.byref(d), _p(gy), _p(x), _p(gwf), None, 1, 0, _st())
        if ctx.has[0] and ctx.needs_input_grad[2]:
            gb = gb_of_split[0] if gb_of_split else _bias_grad_of(gy, cb)
        if ctx.has[1] and ctx.needs_input_grad[3]:
            gres = gy
        return gx, gwf, gb, gres, None, None, None, None, None, None, None


class _ConvTransposePair(torch.autograd.Function):
    """ConvTranspose2d(x1, W1) + ConvTranspose2d(x2, W2) + bias, both kernel 3 / stride 2 / padding 1 / output_padding 1, in ONE launch
    (fmi_conv_transpose2d_pair_f32: ResBlockDecoder's main path and bypass, base_function.py:297-305).  The backward is the two single
    ConvTranspose2d backwards on the shared gradient."""

    @staticmethod
    def forward(ctx, x1, wf1, wt1, x2, wf2, wt2, bias, w3_1, w3_2):
        _chk(x1, wf1, x2, wf2, bias)
        n, h, w, cs1 = x1.shape
        cs2, cb = x2.shape[3], wf1.shape[1]
        H, W = 2 * h, 2 * w
        d, oh, ow = conv_desc(n, H, W, cb, cs1, 3, 3, 2, 1, w3=w3_1[1])
        y = torch.empty((n, H, W, cb), device=x1.device, dtype=torch.float32)
        with _prof(f"convT_pair_fwd|{n}x{h}x{w} {cs1}+{cs2}->{cb}", 2.0 * (x1.numel() + x2.numel()) * cb * 9):
            _L().conv_transpose2d_pair_f32(C.byref(d), _p(x1), _p(x2), cs2, C.c_void_p(w3_2[1].data_ptr()), _p(bias), _p(y), _st())
        ctx.save_for_backward(x1, wf1, x2, wf2)
        ctx.w3s, ctx.HW, ctx.has_bias = (w3_1, w3_2), (H, W), bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        import types

        x1, wf1, x2, wf2 = ctx.saved_tensors
        gy = gy.contiguous()
        outs = []
        for x, wf, w3, ix, iw in ((x1, wf1, ctx.w3s[0], 0, 1), (x2, wf2, ctx.w3s[1], 3, 4)):
            ns = types.SimpleNamespace(saved_tensors=(x, wf), cfg=(3, 3, 2, 1), has=(False, False), HW=ctx.HW, wf3=w3[0], x3=None,
                                       needs_input_grad=(ctx.needs_input_grad[ix], ctx.needs_input_grad[iw]) + (False,) * 9)
            outs.append(_ConvTranspose2d.backward(ns, gy)[:2])
        gb = _bias_grad_of(gy, gy.shape[-1]) if (ctx.has_bias and ctx.needs_input_grad[6]) else None
        return outs[0][0], outs[0][1], None, outs[1][0], outs[1][1], None, gb, None, None


def conv_transpose2d_pair_ok(x1, pw1: PackedWeight, x2, pw2: PackedWeight) -> bool:
    """the shapes fmi_conv_transpose2d_pair_f32 takes: fp32, thin outputs on a large map (convt3x3.h)"""
    if x1.dtype != torch.float32 or x2.dtype != torch.float32 or not x1.is_cuda or x1.shape[:3] != x2.shape[:3]:
        return False
    if os.environ.get("FMI_CT3_OFF"):
        return False
    n, h, w, cs1 = x1.shape
    cs2, cb = x2.shape[3], pw1.wf.shape[1]
    return (pw1.kh == 3 and pw1.kw == 3 and pw2.kh == 3 and pw2.kw == 3 and pw2.wf.shape[1] == cb and cb <= 64 and cb % 4 == 0 and cs1 % 16 == 0 and
            cs2 % 16 == 0 and n * h * w >= 32768 and pw1.w3[1] is not None and pw2.w3[1] is not None and x1.is_contiguous() and x2.is_contiguous() and
            not p3_wanted(n * h * w, cs1, cb, 9))


def conv_transpose2d_pair(x1, pw1: PackedWeight, x2, pw2: PackedWeight, bias=None):
    return _ConvTransposePair.apply(x1, pw1.wf, pw1.wt, x2, pw2.wf, pw2.wt, bias, pw1.w3, pw2.w3)


def conv_transpose2d(x, pw: PackedWeight, bias=None, residual=None, stride=2, pad=1, out_pad=1):
    if x.dtype == BF16:
        if bias is not None or residual is not None:
            raise FmiError("the bf16 ConvTranspose2d has no bias / residual epilogue")
        return _ConvTranspose2dBF16.apply(x, pw.wf, pw.wt, pw.kh, pw.kw, stride, pad, out_pad)
    return _ConvTranspose2d.apply(x, pw.wf, bias, residual, pw.wt, pw.kh, pw.kw, stride, pad, out_pad, pw.w3)


# ---------------------------------------------------------------------------------------------------
# element-wise / pooling / normalisation
# ---------------------------------------------------------------------------------------------------
class _LeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        ctx.save_for_backward(x)
        ctx.slope = slope
        return eltwise(EW_LRELU, x, None, slope)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return eltwise(EW_LRELU_BWD, g.contiguous(), x, ctx.slope), None


def leaky_relu(x, slope=0.1):
    return _LeakyReLU.apply(x, slope)


class _Softplus(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return eltwise(EW_SOFTPLUS, x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return eltwise(EW_SOFTPLUS_BWD, g.contiguous(), x)


class _MulAdd(torch.autograd.Function):
    """a * b + c with b a constant (the re-parameterisation noise)."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(b)
        return eltwise(EW_ADD, eltwise(EW_MUL, a, b), c)

    @staticmethod
    def backward(ctx, g):
        (b,) = ctx.saved_tensors
        g = g.contiguous()
        return eltwise(EW_MUL, g, b), None, g


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.dtype == BF16:
            _chk(a, b, dtype=BF16)
            y = torch.empty_like(a)
            _L().add_bf16(_p(a), _p(b), _p(y), a.numel(), _st())
            return y
        return eltwise(EW_ADD, a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _chk(x)
        n, h, w, c = x.shape
        y = torch.empty((n, h // k, w // k, c), device=x.device, dtype=torch.float32)
        _L().avgpool_f32(_p(x), _p(y), n, h, w, c, k, _st())
        ctx.shape, ctx.k = x.shape, k
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _L().avgpool_bwd_f32(_p(g.contiguous()), _p(gx), n, h, w, c, ctx.k, _st())
        return gx, None


def avg_pool(x, k=2):
    return _AvgPool.apply(x, k)


class _AdaptiveAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        _chk(x)
        n, h, w, c = x.shape
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        _L().adaptive_avgpool_f32(_p(x), _p(y), n, h, w, c, oh, ow, _st())
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _L().adaptive_avgpool_bwd_f32(_p(g.contiguous()), _p(gx), n, h, w, c, g.shape[1], g.shape[2], _st())
        return gx, None, None


def adaptive_avg_pool(x, oh, ow):
    """nn.AdaptiveAvgPool2d((oh, ow)) on NHWC for any sizes; integer down-sampling factors take the k x k kernel"""
    n, h, w, c = x.shape
    if (h, w) == (oh, ow):
        return x
    if h % oh == 0 and w % ow == 0 and h // oh == w // ow:
        return avg_pool(x, h // oh)
    return _AdaptiveAvgPool.apply(x, int(oh), int(ow))


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        _chk(x)
        n, h, w, c = x.shape
        oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        arg = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.int32)
        _L().maxpool_f32(_p(x), _p(y), C.c_void_p(arg.data_ptr()), n, h, w, c, k, stride, _st())
        ctx.save_for_backward(arg)
        ctx.cfg = (x.shape, k, stride)
        return y

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        (n, h, w, c), k, stride = ctx.cfg
        gx = torch.zeros((n, h, w, c), device=g.device, dtype=torch.float32)
        _L().maxpool_bwd_f32(_p(g.contiguous()), C.c_void_p(arg.data_ptr()), _p(gx), n, h, w, c, k, stride, _st())
        return gx, None, None


def max_pool(x, k, stride):
    """nn.MaxPool2d(k, stride) (no padding, floor mode) on NHWC"""
    if k == 2 and stride == 2 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0:
        return max_pool2(x)
    return _MaxPool.apply(x, int(k), int(stride))


def argmax_channels(x_nhwc: torch.Tensor) -> torch.Tensor:
    """x [N,H,W,C] -> float mask [N,H,W] = argmax over C (first maximum wins): mask_detector(...).argmax(1).float(),
    PICNet_inference.py:100-101; index work, bit exact, no gradient"""
    x = x_nhwc.detach()
    _chk(x)
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    _L().argmax_channels_f32(_p(x), _p(out), out.numel(), x.shape[-1], _st())
    return out


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        n, h, w, c = x.shape
        y = torch.empty((n, h // 2, w // 2, c), device=x.device, dtype=torch.float32)
        _L().maxpool2_f32(_p(x), _p(y), n, h, w, c, _st())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        gx = torch.empty_like(x)
        _L().maxpool2_bwd_f32(_p(x), _p(g.contiguous()), _p(gx), n, h, w, c, _st())
        return gx


def max_pool2(x):
    return _MaxPool2.apply(x)


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow, mean, std):
        _chk(x, mean, std)
        n, h, w, c = x.shape
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        _L().resize_bilinear_f32(_p(x), _p(y), n, h, w, c, oh, ow, _p(mean), _p(std), _st())
        ctx.shape, ctx.std = x.shape, std
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        gx = _zeros(ctx.shape, g.device, torch.float32)
        _L().resize_bilinear_bwd_f32(_p(g.contiguous()), _p(gx), n, h, w, c, g.shape[1], g.shape[2], _p(ctx.std), _st())
        return gx, None, None, None, None


def resize_bilinear(x, oh, ow, mean=None, std=None):
    """bilinear, align_corners=True, optionally followed by (v - mean[c]) / std[c]."""
    return _Resize.apply(x, oh, ow, mean, std)


def _norm_ws(device, n, c):
    """(sums [n, c, 2] fp64 -- written by the library --, partials workspace): the statistics reductions store one partial row per
    workgroup and add the rows in a second launch instead of serialising fp64 atomics"""
    sums = torch.empty((n, c, 2), device=device, dtype=torch.float64)
    ws = torch.empty(max(1024, n) * c * 2, device=device, dtype=torch.float64)
    return sums, ws


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, slope, passthrough=False):
        _chk(x, gamma, beta)
        lib = _L()
        n, h, w, c = x.shape
        sums, ws = _norm_ws(x.device, n, c)
        stats = torch.empty((n, c, 2), device=x.device, dtype=torch.float32)
        lib.instnorm_stats_f32(_p(x), C.c_void_p(sums.data_ptr()), _p(stats), n, h * w, c, eps, C.c_void_p(ws.data_ptr()), ws.numel(), _st())
        y = torch.empty_like(x)
        lib.instnorm_apply_f32(_p(x), _p(stats), _p(gamma), _p(beta), _p(y), n, h * w, c, slope, _st())
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.slope = slope
        if passthrough:  # x for its other consumer (the bypass ConvTranspose2d of a ResBlockDecoder): its gradient re-enters below
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, g, gpass=None):
        lib = _L()
        x, stats, gamma, beta = ctx.saved_tensors
        if g is None:
            return gpass, None, None, None, None, None
        n, h, w, c = x.shape
        g = g.contiguous()
        red, ws = _norm_ws(x.device, n, c)
        lib.instnorm_bwd_reduce_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), n, h * w, c, ctx.slope,
                                    C.c_void_p(ws.data_ptr()), ws.numel(), _st())
        gx = torch.empty_like(x)
        dg = _zeros_like(gamma)
        db = _zeros_like(beta)
        if gpass is not None:
            lib.instnorm_bwd_apply_add_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), _p(gpass.contiguous()),
                                           _p(gx), _p(dg), _p(db), n, h * w, c, ctx.slope, _st())
        else:
            lib.instnorm_bwd_apply_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), _p(gx), _p(dg), _p(db),
                                       n, h * w, c, ctx.slope, _st())
        return gx, dg, db, None, None, None


def instance_norm_act(x, gamma, beta, eps=1e-5, slope=1.0, passthrough=False):
    """lrelu(InstanceNorm2d(affine)(x)); slope = 1 disables the activation.  passthrough: returns (y, x') -- see _InstNormAct"""
    return _InstNormAct.apply(x, gamma, beta, eps, slope, passthrough)


class _MaskMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, invert):
        _chk(x, m)
        y = torch.empty_like(x)
        c = x.shape[-1]
        _L().mask_mul_f32(_p(x), _p(m), _p(y), x.numel() // c, c, invert, _st())
        ctx.save_for_backward(m)
        ctx.invert = invert
        return y

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(g)
        c = g.shape[-1]
        _L().mask_mul_f32(_p(g), _p(m), _p(gx), g.numel() // c, c, ctx.invert, _st())
        return gx, None, None


def mask_mul(x, m, invert=False):
    """x[N,H,W,C] * m[N,H,W] (or 1-m)."""
    return _MaskMul.apply(x, m, 1 if invert else 0)


class _SliceChannels(torch.autograd.Function):
    """x[..., c0:c0 + c] as a contiguous tensor (strided copy kernel); the gradient is written into a zeroed full-width tensor"""

    @staticmethod
    def forward(ctx, x, c0, c):
        _chk(x)
        ctot = x.shape[-1]
        y = torch.empty(x.shape[:-1] + (c,), device=x.device, dtype=torch.float32)
        _L().copy_channels_f32(_p(x), _p(y), x.numel() // ctot, ctot, c0, c, 0, c, _st())
        ctx.cfg = (x.shape, c0, c)
        return y

    @staticmethod
    def backward(ctx, g):
        shape, c0, c = ctx.cfg
        gx = torch.zeros(shape, device=g.device, dtype=torch.float32)
        _L().copy_channels_f32(_p(g.contiguous()), _p(gx), g.numel() // c, c, 0, shape[-1], c0, c, _st())
        return gx, None, None


def slice_channels(x, c0, c):
    return _SliceChannels.apply(x, int(c0), int(c))


class _CatChannels(torch.autograd.Function):
    """torch.cat([a, b], dim=-1) of two NHWC maps as two strided copies"""

    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        ca, cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // ca
        y = torch.empty(a.shape[:-1] + (ca + cb,), device=a.device, dtype=torch.float32)
        _L().copy_channels_f32(_p(a), _p(y), rows, ca, 0, ca + cb, 0, ca, _st())
        _L().copy_channels_f32(_p(b), _p(y), rows, cb, 0, ca + cb, ca, cb, _st())
        ctx.cfg = (ca, cb)
        return y

    @staticmethod
    def backward(ctx, g):
        ca, cb = ctx.cfg
        g = g.contiguous()
        rows = g.numel() // (ca + cb)
        ga = torch.empty(g.shape[:-1] + (ca,), device=g.device, dtype=torch.float32)
        gb = torch.empty(g.shape[:-1] + (cb,), device=g.device, dtype=torch.float32)
        _L().copy_channels_f32(_p(g), _p(ga), rows, ca + cb, 0, ca, 0, ca, _st())
        _L().copy_channels_f32(_p(g), _p(gb), rows, ca + cb, ca, cb, 0, cb, _st())
        return ga, gb


def cat_channels(a, b):
    return _CatChannels.apply(a.contiguous(), b.contiguous())


def binarise_mask(mask_i64: torch.Tensor) -> torch.Tensor:
    """(mask > 0).float() -- train_reference_fill.py:340; bit exact."""
    if not mask_i64.is_cuda or mask_i64.dtype != torch.int64:
        raise FmiError("binarise_mask needs an int64 device tensor")
    m = mask_i64.contiguous()
    out = torch.empty(m.shape, device=m.device, dtype=torch.float32)
    _L().mask_binarise_i64(C.c_void_p(m.data_ptr()), _p(out), m.numel(), _st())
    return out


class _GuideBlend(torch.autograd.Function):
    """out[..., :C] = (1-m) ref_att + m ref ; out[..., C:] = src_att   (example_guided_att.py:35-37)."""

    @staticmethod
    def forward(ctx, ref_att, ref, src_att, m):
        _chk(ref_att, ref, src_att, m)
        n, h, w, c = ref.shape
        out = torch.empty((n, h, w, 2 * c), device=ref.device, dtype=torch.float32)
        _L().guide_blend_f32(_p(ref_att), _p(ref), _p(m), _p(out), n * h * w, c, 2 * c, _st())
        out[..., c:] = src_att
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = g.contiguous()
        n, h, w, c2 = g.shape
        c = c2 // 2
        gra = torch.empty((n, h, w, c), device=g.device, dtype=torch.float32)
        grf = torch.empty_like(gra)
        _L().guide_blend_bwd_f32(_p(g), _p(m), _p(gra), _p(grf), n * h * w, c, c2, _st())
        return gra, grf, g[..., c:].contiguous(), None


def guide_blend_cat(ref_att, ref, src_att, m):
    return _GuideBlend.apply(ref_att, ref, src_att, m)


class _ScaleAddParam(torch.autograd.Function):
    """y = gamma * o + x with gamma a 1-element device parameter (base_function.py:439)."""

    @staticmethod
    def forward(ctx, o, gamma, x):
        _chk(o, gamma, x)
        y = torch.empty_like(x)
        _L().axpy_dev_f32(_p(o), _p(gamma), _p(x), _p(y), x.numel(), _st())
        ctx.save_for_backward(o, gamma)
        return y

    @staticmethod
    def backward(ctx, g):
        o, gamma = ctx.saved_tensors
        g = g.contiguous()
        go = torch.empty_like(g)
        _L().axpy_dev_f32(_p(g), _p(gamma), None, _p(go), g.numel(), _st())
        gg = _zeros_like(gamma)
        _L().dot_f32(_p(g), _p(o), g.numel(), 1.0, _p(gg), _st())
        return go, gg, g


def scale_add_param(o, gamma, x):
    return _ScaleAddParam.apply(o, gamma, x)


class _VaeSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o_src, o_ref, eps_q, eps_p):
        _chk(o_src, o_ref, eps_q, eps_p)
        n, h, w, c2 = o_src.shape
        z = torch.empty_like(o_src)
        _L().vae_sample_f32(_p(o_src), _p(o_ref), _p(eps_q), _p(eps_p), _p(z), n * h * w, c2 // 2, _st())
        ctx.save_for_backward(o_src, o_ref, eps_q, eps_p)
        return z

    @staticmethod
    def backward(ctx, g):
        o_src, o_ref, eps_q, eps_p = ctx.saved_tensors
        n, h, w, c2 = o_src.shape
        gs, gr = torch.empty_like(o_src), torch.empty_like(o_ref)
        _L().vae_sample_bwd_f32(_p(g.contiguous()), _p(o_src), _p(o_ref), _p(eps_q), _p(eps_p), _p(gs), _p(gr), n * h * w, c2 // 2, _st())
        return gs, gr, None, None


def vae_sample(o_src, o_ref, eps_q, eps_p):
    """z = cat[mu_q + softplus(s_q) eps_q, mu_p + softplus(s_p) eps_p] (network.py:167-168,275-307)."""
    return _VaeSample.apply(o_src, o_ref, eps_q, eps_p)


# ---------------------------------------------------------------------------------------------------
# attention:  A = softmax(Q Q^T) ;  O_i = A V_i        (Q [N,T,d], V_i [N,T,C_i])
# chunked over query rows so that one score chunk stays inside the Infinity Cache
# ---------------------------------------------------------------------------------------------------
def _attn_plan(n, t, budget):
    """(images per group, query rows per chunk): long reductions for the P^T dO / dS^T Q products want many rows per
    chunk, so images are processed in groups rather than all at once with 128-row chunks"""
    qc = min(t, 512)
    if qc * t * 4 > budget:
        qc = max(128, (budget // (t * 4) // 128) * 128)
        qc = min(t, qc)
    ng = max(1, min(n, budget // (qc * t * 4)))
    return ng, qc


def _scores(q, ng, t, d, n0, q0, qc, buf):
    # S[b, i, j] = q[n0+b, q0+i, :] . q[n0+b, j, :]
    gemm_raw(_p(q, (n0 * t + q0) * d), _p(q, n0 * t * d), _p(buf), qc, t, d, (d, 1), (1, d), (t, 1), ng, (t * d, t * d, qc * t), tag="attn_qk")
    _L().softmax_rows_f32(_p(buf), _p(buf), ng * qc, t, _st())


FUSED_ATTENTION = True  # tests flip this to cross-check the fused kernel against the GEMM + softmax composition


def _fused_attn_ok(q, vs):
    n, t, d = q.shape
    if not FUSED_ATTENTION or len(vs) > 2 or t % 128 or d not in (16, 32, 64):
        return False
    cs = [v.shape[2] for v in vs]
    if any(c % 32 for c in cs):
        return False
    return (d, sum(cs) // 32) in ((64, 8), (32, 8), (32, 4), (64, 4), (16, 2), (16, 4))


class _SelfAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, *vs):
        _chk(q, *vs)
        n, t, d = q.shape
        if _fused_attn_ok(q, vs):
            outs = [torch.empty_like(v) for v in vs]
            lse = torch.empty((n, t), device=q.device, dtype=torch.float32)
            c1 = vs[0].shape[2]
            c2 = vs[1].shape[2] if len(vs) > 1 else 0
            with _prof(f"attn_fused_fwd|T{t} d{d} C{c1 + c2} b{n}", 2.0 * n * t * t * (d + c1 + c2)):
                _L().attention_fwd_f32(_p(q), _p(vs[0]), _p(vs[1]) if len(vs) > 1 else None, _p(outs[0]),
                                       _p(outs[1]) if len(vs) > 1 else None, _p(lse), n, t, d, c1, c2, _st())
            # the outputs go through save_for_backward: as plain ctx attributes they would close a reference cycle through their
            # own grad_fn (node -> ctx -> output -> node) that Python's collector cannot see -- one whole discriminator graph leaked
            # per training step (150 MB)
            ctx.save_for_backward(q, *vs, lse, *outs)
            ctx.nv = len(vs)
            return tuple(outs)
        ngm, qcm = _attn_plan(n, t, ATTN_CHUNK_BYTES)
        buf = torch.empty(ngm * qcm * t, device=q.device, dtype=torch.float32)
        outs = [torch.empty_like(v) for v in vs]
        for n0 in range(0, n, ngm):
            ng = min(ngm, n - n0)
            for q0 in range(0, t, qcm):
                qc = min(qcm, t - q0)
                _scores(q, ng, t, d, n0, q0, qc, buf)
                for v, o in zip(vs, outs):
                    c = v.shape[2]
                    gemm_raw(_p(buf), _p(v, n0 * t * c), _p(o, (n0 * t + q0) * c), qc, c, t, (t, 1), (c, 1), (c, 1), ng,
                             (qc * t, t * c, t * c), tag="attn_pv")
        ctx.save_for_backward(q, *vs)
        ctx.nv = -1
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gos):
        if ctx.nv > 0:
            q, *rest = ctx.saved_tensors
            vs, lse, fwd_outs = rest[:ctx.nv], rest[ctx.nv], rest[ctx.nv + 1:]
        else:
            q, *vs = ctx.saved_tensors
            lse, fwd_outs = None, None
        n, t, d = q.shape
        csum = sum(v.shape[2] for v in vs)
        if lse is not None and FUSED_ATTENTION and (d, csum // 32) in ((64, 8), (32, 8), (32, 4), (64, 4)) and all(g is not None for g in gos):
            gos = [g.contiguous() for g in gos]
            gq = _zeros_like(q)
            gvs = [torch.empty_like(v) for v in vs]
            delta = torch.empty((n, t), device=q.device, dtype=torch.float32)
            c1 = vs[0].shape[2]
            c2 = vs[1].shape[2] if len(vs) > 1 else 0
            two = len(vs) > 1
            with _prof(f"attn_fused_bwd|T{t} d{d} C{c1 + c2} b{n}", 2.0 * n * t * t * (3 * d + 2 * (c1 + c2))):
                _L().attention_bwd_f32(_p(q), _p(vs[0]), _p(vs[1]) if two else None, _p(fwd_outs[0]), _p(fwd_outs[1]) if two else None,
                                       _p(gos[0]), _p(gos[1]) if two else None, _p(lse), _p(delta), _p(gvs[0]), _p(gvs[1]) if two else None,
                                       _p(gq), n, t, d, c1, c2, _st())
            return (gq,) + tuple(gvs)
        ngm, qcm = _attn_plan(n, t, ATTN_CHUNK_BYTES // 2)
        P = torch.empty(ngm * qcm * t, device=q.device, dtype=torch.float32)
        dP = torch.empty(ngm * qcm * t, device=q.device, dtype=torch.float32)
        gos = [g.contiguous() if g is not None else None for g in gos]
        gq = _zeros_like(q)
        gvs = [_zeros_like(v) if g is not None else None for v, g in zip(vs, gos)]
        if all(g is None for g in gos):
            return (gq,) + tuple(gvs)
        for n0 in range(0, n, ngm):
            ng = min(ngm, n - n0)
            for q0 in range(0, t, qcm):
                qc = min(qcm, t - q0)
                _scores(q, ng, t, d, n0, q0, qc, P)
                first = True
                for v, g, gv in zip(vs, gos, gvs):
                    if g is None:
                        continue
                    c = v.shape[2]
                    # dV += P^T dO_chunk
                    gemm_raw(_p(P), _p(g, (n0 * t + q0) * c), _p(gv, n0 * t * c), t, c, qc, (1, t), (c, 1), (c, 1), ng,
                             (qc * t, t * c, t * c), 1.0, 1.0, tag="attn_bwd_dv")
                    # dP (+)= dO_chunk V^T
                    gemm_raw(_p(g, (n0 * t + q0) * c), _p(v, n0 * t * c), _p(dP), qc, t, c, (c, 1), (1, c), (t, 1), ng,
                             (t * c, t * c, qc * t), 1.0, 0.0 if first else 1.0, tag="attn_bwd_dp")
                    first = False
                _L().softmax_rows_bwd_f32(_p(P), _p(dP), _p(dP), ng * qc, t, _st())
                # query side: dQ[chunk] += dS Q ; key side: dQ += dS^T Q[chunk]
                gemm_raw(_p(dP), _p(q, n0 * t * d), _p(gq, (n0 * t + q0) * d), qc, d, t, (t, 1), (d, 1), (d, 1), ng,
                         (qc * t, t * d, t * d), 1.0, 1.0, tag="attn_bwd_dq")
                gemm_raw(_p(dP), _p(q, (n0 * t + q0) * d), _p(gq, n0 * t * d), t, d, qc, (1, t), (d, 1), (d, 1), ng,
                         (qc * t, t * d, t * d), 1.0, 1.0, tag="attn_bwd_dk")
        return (gq,) + tuple(gvs)


def self_attention(q: torch.Tensor, values: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, ...]:
    """q [N,T,d]; values: list of [N,T,C]; returns softmax(q q^T) @ v for each v."""
    return _SelfAttention.apply(q, *values)


# ---------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------
class _ReduceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, a, b, c0, scale):
        _chk(a, b)
        out = _zeros((), a.device, torch.float32)
        _L().reduce_loss_f32(kind, _p(a), _p(b), a.numel(), c0, scale, _p(out), _st())
        ctx.save_for_backward(a, b)
        ctx.cfg = (kind, c0, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        kind, c0, scale = ctx.cfg
        ga = torch.empty_like(a)
        _L().reduce_loss_bwd_f32(kind, _p(a), _p(b), a.numel(), c0, scale, _p(g.contiguous()), _p(ga), _st())
        gb = None
        if b is not None and ctx.needs_input_grad[2]:
            gb = eltwise(EW_SCALE, ga, None, -1.0)
        return None, ga, gb, None, None


def l1_loss(a, b):
    """mean |a - b| (nn.L1Loss / F.l1_loss)."""
    return _ReduceLoss.apply(0, a.contiguous(), b.contiguous(), 0.0, 1.0 / a.numel())


def mse_loss(a, b):
    return _ReduceLoss.apply(1, a.contiguous(), b.contiguous(), 0.0, 1.0 / a.numel())


def mse_to_const(a, c0):
    """mean (a - c0)^2: LSGAN objective against a constant label (external_function.py:113-115)."""
    return _ReduceLoss.apply(2, a.contiguous(), None, float(c0), 1.0 / a.numel())


class _L2NormRows(torch.autograd.Function):
    """y = x / (||x|| + eps) over the last dimension"""

    @staticmethod
    def forward(ctx, x, eps):
        _chk(x)
        c = x.shape[-1]
        rows = x.numel() // c
        y = torch.empty_like(x)
        inv = torch.empty(rows, device=x.device, dtype=torch.float32)
        _L().l2norm_rows_f32(_p(x), _p(y), _p(inv), rows, c, eps, _st())
        ctx.save_for_backward(y, inv)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        c = y.shape[-1]
        gx = torch.empty_like(y)
        _L().l2norm_rows_bwd_f32(_p(g.contiguous()), _p(y), _p(inv), _p(gx), y.numel() // c, c, ctx.eps, _st())
        return gx, None


def l2norm_rows(x, eps=0.0):
    return _L2NormRows.apply(x.contiguous(), float(eps))


class _LpipsLayer(torch.autograd.Function):
    """scale * sum_p sum_c w[c] (fx - fy)^2 for NHWC feature maps fx, fy [N, H, W, C] (one LPIPS layer, lpips.py:33-36)"""

    @staticmethod
    def forward(ctx, fx, fy, w, scale):
        _chk(fx, fy, w)
        c = fx.shape[-1]
        out = _zeros((), fx.device, torch.float32)
        _L().lpips_layer_f32(_p(fx), _p(fy), _p(w), _p(out), fx.numel() // c, c, scale, _st())
        ctx.save_for_backward(fx, fy, w)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        fx, fy, w = ctx.saved_tensors
        c = fx.shape[-1]
        gfx = torch.empty_like(fx) if ctx.needs_input_grad[0] else None
        gfy = torch.empty_like(fy) if ctx.needs_input_grad[1] else None
        if gfx is not None or gfy is not None:
            _L().lpips_layer_bwd_f32(_p(fx), _p(fy), _p(w), _p(g.contiguous()), _p(gfx), _p(gfy), fx.numel() // c, c, ctx.scale, _st())
        return gfx, gfy, None, None


def lpips_layer(fx, fy, w, scale):
    return _LpipsLayer.apply(fx.contiguous(), fy.contiguous(), w.contiguous(), float(scale))


class _DotAll(torch.autograd.Function):
    """scale * sum(a * b) over all elements"""

    @staticmethod
    def forward(ctx, a, b, scale):
        _chk(a, b)
        out = _zeros((), a.device, torch.float32)
        _L().dot_f32(_p(a), _p(b), a.numel(), scale, _p(out), _st())
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        k = g.contiguous().view(1) * ctx.scale
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            _L().axpy_dev_f32(_p(b), _p(k), None, _p(ga), a.numel(), _st())
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            _L().axpy_dev_f32(_p(a), _p(k), None, _p(gb), a.numel(), _st())
        return ga, gb, None


def dot_all(a, b, scale=1.0):
    return _DotAll.apply(a.contiguous(), b.contiguous(), float(scale))


class _Gram(torch.autograd.Function):
    """G[n] = X[n]^T X[n] / (C*P), X [N,P,C]   (external_function.py:180-185)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        n, p, c = x.shape
        g = torch.empty((n, c, c), device=x.device, dtype=torch.float32)
        gemm_raw(_p(x), _p(x), _p(g), c, c, p, (1, c), (c, 1), (c, 1), n, (p * c, p * c, c * c), 1.0 / (c * p))
        ctx.save_for_backward(x)
        return g

    @staticmethod
    def backward(ctx, dg):
        (x,) = ctx.saved_tensors
        n, p, c = x.shape
        dg = dg.contiguous()
        gx = torch.empty_like(x)
        a = 1.0 / (c * p)
        # dX = a * X (dG + dG^T)
        gemm_raw(_p(x), _p(dg), _p(gx), p, c, c, (c, 1), (c, 1), (c, 1), n, (p * c, c * c, p * c), a, 0.0)
        gemm_raw(_p(x), _p(dg), _p(gx), p, c, c, (c, 1), (1, c), (c, 1), n, (p * c, c * c, p * c), a, 1.0)
        return gx


def gram_matrix(x_npc):
    return _Gram.apply(x_npc)


class _ContextualLoss(torch.autograd.Function):
    """external_function.py:231-274 on NHWC features flattened to [N,P,C]; y carries no gradient."""

    @staticmethod
    def forward(ctx, x, y, h, scale):
        _chk(x, y)
        lib = _L()
        n, p, c = x.shape
        dev = x.device
        mu = _zeros(c, dev, torch.float32)
        lib.cx_channel_mean_f32(_p(y), _p(mu), n * p, c, _st())
        xn, yn = torch.empty_like(x), torch.empty_like(y)
        xi, yi = torch.empty(n * p, device=dev), torch.empty(n * p, device=dev)
        lib.cx_normalise_f32(_p(x), _p(mu), _p(xn), _p(xi), n * p, c, _st())
        lib.cx_normalise_f32(_p(y), _p(mu), _p(yn), _p(yi), n * p, c, _st())
        cosm = torch.empty((n, p, p), device=dev, dtype=torch.float32)
        gemm_raw(_p(xn), _p(yn), _p(cosm), p, p, c, (c, 1), (1, c), (p, 1), n, (p * c, p * c, p * p))
        cxij = torch.empty_like(cosm)
        dmin = torch.empty(n * p, device=dev)
        amin = torch.empty(n * p, device=dev, dtype=torch.int32)
        rsum = torch.empty(n * p, device=dev)
        lib.cx_rows_f32(_p(cosm), _p(cxij), _p(dmin), C.c_void_p(amin.data_ptr()), _p(rsum), n, p, h, _st())
        cmax = torch.empty(n * p, device=dev)
        carg = torch.empty(n * p, device=dev, dtype=torch.int32)
        lib.cx_cols_f32(_p(cxij), _p(cmax), C.c_void_p(carg.data_ptr()), n, p, _st())
        cx = torch.empty(n, device=dev)
        loss = _zeros((), dev, torch.float32)
        lib.cx_loss_f32(_p(cmax), _p(cx), _p(loss), n, p, scale, _st())
        ctx.save_for_backward(xn, yn, xi, cosm, cxij, dmin, amin, rsum, carg, cx)
        ctx.cfg = (h, scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _L()
        xn, yn, xi, cosm, cxij, dmin, amin, rsum, carg, cx = ctx.saved_tensors
        h, scale = ctx.cfg
        n, p, c = xn.shape
        dcos = torch.empty_like(cosm)
        lib.cx_bwd_f32(_p(cxij), _p(dmin), C.c_void_p(amin.data_ptr()), _p(rsum), _p(cosm), C.c_void_p(carg.data_ptr()), _p(cx),
                       _p(g.contiguous()), _p(dcos), n, p, h, scale, _st())
        gxn = torch.empty_like(xn)
        gemm_raw(_p(dcos), _p(yn), _p(gxn), p, c, p, (p, 1), (c, 1), (c, 1), n, (p * p, p * c, p * c))
        gx = torch.empty_like(xn)
        lib.cx_normalise_bwd_f32(_p(gxn), _p(xn), _p(xi), _p(gx), n * p, c, _st())
        return gx, None, None, None


def contextual_loss(x_npc, y_npc, h=0.5, scale=1.0):
    return _ContextualLoss.apply(x_npc.contiguous(), y_npc.contiguous(), h, scale)


# ---------------------------------------------------------------------------------------------------
# StyleGAN2 decoder pieces (NHWC)
# ---------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = alpha * x W^T + bias   (x [N,K], W [O,K]) -- EqualLinear, stylegan2/model.py:160-165"""

    @staticmethod
    def forward(ctx, x, w, bias, alpha):
        _chk(x, w, bias)
        n, k = x.shape
        o = w.shape[0]
        y = torch.empty((n, o), device=x.device, dtype=torch.float32)
        gemm_raw(_p(x), _p(w), _p(y), n, o, k, (k, 1), (1, k), (o, 1), alpha=alpha, bias=bias, tag="linear")
        ctx.save_for_backward(x, w)
        ctx.alpha, ctx.has_bias = alpha, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        n, k = x.shape
        o = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            gemm_raw(_p(g), _p(w), _p(gx), n, k, o, (o, 1), (k, 1), (k, 1), alpha=ctx.alpha, tag="linear_bwd")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            gemm_raw(_p(g), _p(x), _p(gw), o, k, n, (1, o), (k, 1), (k, 1), alpha=ctx.alpha, tag="linear_bwd")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _zeros(o, g.device, torch.float32)
            _L().bias_grad_f32(_p(g), n, o, o, _p(gb), _st())
        return gx, gw, gb, None


def linear(x, w, bias=None, alpha=1.0):
    return _Linear.apply(x.contiguous(), w.contiguous(), bias, float(alpha))


def _scale_bwd_ok(g, x, s, c):
    """shapes the one-pass adjoint (fmi_scale_channels_bwd_f32) takes; off with FMI_SCALE_BWD_FUSED=0 (A/B)"""
    return (_SCALE_BWD_FUSED and g.dtype == torch.float32 and c % 4 == 0 and c <= 1024 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and
            s.data_ptr() % 16 == 0)


def _scale_channels_bwd(g, x, s, n, p, c):
    gx, gs = torch.empty_like(x), torch.empty_like(s)
    ws = _parts_ws(x.device, max(4096, n) * c)
    _L().scale_channels_bwd_f32(_p(g), _p(x), _p(s), _p(gx), _p(gs), _p(ws), ws.numel(), n, p, c, _st())
    return gx, gs


_SCALE_BWD_FUSED = os.environ.get("FMI_SCALE_BWD_FUSED", "1") != "0"


class _ScaleChannels(torch.autograd.Function):
    """y[n,p,c] = x[n,p,c] * s[n,c]"""

    @staticmethod
    def forward(ctx, x, s):
        _chk(x, dtype=x.dtype)
        _chk(s)
        n, c = s.shape
        p = x.numel() // (n * c)
        y = torch.empty_like(x)
        (_L().scale_channels_bf16 if x.dtype == BF16 else _L().scale_channels_f32)(_p(x), _p(s), _p(y), n, p, c, _st())
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = g.contiguous()
        n, c = s.shape
        p = x.numel() // (n * c)
        gx = gs = None
        b16 = x.dtype == BF16
        if not b16 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _scale_bwd_ok(g, x, s, c):
            return _scale_channels_bwd(g, x, s, n, p, c)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            (_L().scale_channels_bf16 if b16 else _L().scale_channels_f32)(_p(g), _p(s), _p(gx), n, p, c, _st())
        if ctx.needs_input_grad[1] and b16:
            gs = torch.empty_like(s)
            ws = _parts_ws(x.device, max(2048, n) * c)
            _L().scale_channels_gs_bf16(_p(g), _p(x), _p(gs), _p(ws), ws.numel(), n, p, c, _st())
        elif ctx.needs_input_grad[1]:
            gs = torch.empty_like(s)
            ws = _parts_ws(x.device, max(2048, n) * c)
            _L().scale_channels_gs_f32(_p(g), _p(x), _p(gs), _p(ws), ws.numel(), n, p, c, _st())
        return gx, gs


def scale_channels(x, s):
    return _ScaleChannels.apply(x, s.contiguous())


class _ScaleChannelsAdd(torch.autograd.Function):
    """y[n,p,c] = x[n,p,c] * s[n,c] + res[n,p,c] (fp32): the SE gate and the residual add of an IR-SE block in one pass"""

    @staticmethod
    def forward(ctx, x, s, res):
        _chk(x, s, res)
        n, c = s.shape
        p = x.numel() // (n * c)
        y = torch.empty_like(x)
        _L().scale_channels_add_f32(_p(x), _p(s), _p(res), _p(y), n, p, c, _st())
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = g.contiguous()
        n, c = s.shape
        p = x.numel() // (n * c)
        gx = gs = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _scale_bwd_ok(g, x, s, c):
            gx, gs = _scale_channels_bwd(g, x, s, n, p, c)
            return gx, gs, (g if ctx.needs_input_grad[2] else None)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _L().scale_channels_f32(_p(g), _p(s), _p(gx), n, p, c, _st())
        if ctx.needs_input_grad[1]:
            gs = torch.empty_like(s)
            ws = _parts_ws(x.device, max(2048, n) * c)
            _L().scale_channels_gs_f32(_p(g), _p(x), _p(gs), _p(ws), ws.numel(), n, p, c, _st())
        return gx, gs, (g if ctx.needs_input_grad[2] else None)


class _ScaleChannelsAddBF16(torch.autograd.Function):
    """bf16 twin of _ScaleChannelsAdd (fp32 gates)"""

    @staticmethod
    def forward(ctx, x, s, res):
        _chk(x, res, dtype=BF16)
        _chk(s)
        n, c = s.shape
        p = x.numel() // (n * c)
        y = torch.empty_like(x)
        _L().scale_channels_add_bf16(_p(x), _p(s), _p(res), _p(y), n, p, c, _st())
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = g.contiguous()
        n, c = s.shape
        p = x.numel() // (n * c)
        gx = torch.empty_like(x)
        _L().scale_channels_bf16(_p(g), _p(s), _p(gx), n, p, c, _st())
        gs = torch.empty_like(s)
        ws = _parts_ws(x.device, max(2048, n) * c)
        _L().scale_channels_gs_bf16(_p(g), _p(x), _p(gs), _p(ws), ws.numel(), n, p, c, _st())
        return gx, gs, g


def scale_channels_add(x, s, res):
    if x.dtype == BF16 and x.shape == res.shape and x.shape[-1] % 8 == 0:
        return _ScaleChannelsAddBF16.apply(x.contiguous(), s.contiguous(), res.contiguous())
    if x.dtype != torch.float32 or x.shape[-1] % 4 or x.shape != res.shape:
        return add(scale_channels(x, s), res)
    return _ScaleChannelsAdd.apply(x.contiguous(), s.contiguous(), res.contiguous())


class _GlobalAvgPoolPassBF16(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) of a bf16 NHWC map -> (pooled fp32 [N, C], x') where x' is x for its other consumer (the gated product of
    the SE module); the two gradients meet in ONE pass: gx = g_x' + g_pooled[n][c] / (H W)"""

    @staticmethod
    def forward(ctx, x):
        _chk(x, dtype=BF16)
        n, h, w, c = x.shape
        pooled = torch.empty((n, c), device=x.device, dtype=torch.float32)
        ws = _parts_ws(x.device, 64 * n * c)
        _L().global_avgpool_bf16(_p(x), _p(pooled), _p(ws), ws.numel(), n, h * w, c, _st())
        ctx.shape = x.shape
        return pooled, x.view_as(x)

    @staticmethod
    def backward(ctx, gp, gx_pass):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, device=gp.device, dtype=BF16)
        _L().add_bcast_bf16(_p(gx_pass.contiguous()), _p(gp.contiguous()), _p(gx), n, h * w, c, _st())
        return gx


def global_avg_pool_pass_bf16(x):
    return _GlobalAvgPoolPassBF16.apply(x.contiguous())


class _GlobalAvgPoolPassF32(torch.autograd.Function):
    """fp32 twin of _GlobalAvgPoolPassBF16: AdaptiveAvgPool2d(1) of a square NHWC map -> (pooled [N,1,1,C], x'), x' being x for its
    other consumer; backward: gx = g_x' + g_pooled[n][c] / (H W) in one pass (fmi_add_bcast_f32)"""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        n, h, w, c = x.shape
        y = torch.empty((n, 1, 1, c), device=x.device, dtype=torch.float32)
        _L().avgpool_f32(_p(x), _p(y), n, h, w, c, h, _st())
        ctx.shape = x.shape
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gp, gx_pass):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, device=gp.device, dtype=torch.float32)
        _L().add_bcast_f32(_p(gx_pass.contiguous()), _p(gp.contiguous()), _p(gx), n, h * w, c, _st())
        return gx


def global_avg_pool_pass(x):
    """(pooled [N,1,1,C], x') for a square fp32 NHWC map with C % 4 == 0; see _GlobalAvgPoolPassF32"""
    return _GlobalAvgPoolPassF32.apply(x.contiguous())


class _SqSumLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        rows, k = x.shape
        out = torch.empty(rows, device=x.device, dtype=torch.float32)
        _L().sqsum_last_f32(_p(x), _p(out), rows, k, _st())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        _L().sqsum_last_bwd_f32(_p(x), _p(g.contiguous()), _p(gx), x.shape[0], x.shape[1], _st())
        return gx


def sqsum_last(x2d):
    """sum of squares over the last dimension of a [rows, k] tensor"""
    return _SqSumLast.apply(x2d.contiguous())


class _RsqrtEps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        y = eltwise(EW_RSQRT, x, None, eps)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return eltwise(EW_RSQRT_BWD, g.contiguous(), y), None


def rsqrt_eps(x, eps):
    return _RsqrtEps.apply(x.contiguous(), float(eps))


class _Scale(torch.autograd.Function):
    """y = c * a for a python constant c"""

    @staticmethod
    def forward(ctx, a, c):
        ctx.c = c
        return eltwise(EW_SCALE, a.contiguous(), None, c)

    @staticmethod
    def backward(ctx, g):
        return eltwise(EW_SCALE, g.contiguous(), None, ctx.c), None


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return eltwise(EW_MUL, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        return eltwise(EW_MUL, g, b), eltwise(EW_MUL, g, a)


def mul(a, b):
    return _Mul.apply(a.contiguous(), b.contiguous())


class _NoiseBiasAct(torch.autograd.Function):
    """y = lrelu(x + bias[c] + nw * noise[p], alpha) * scale   (NoiseInjection + FusedLeakyReLU, model.py:282-346)"""

    @staticmethod
    def forward(ctx, x, bias, noise, nw, alpha, scale):
        _chk(x, dtype=x.dtype)
        _chk(bias, noise, nw)
        c = x.shape[-1]
        y = torch.empty_like(x)
        with _prof(f"bytes:noise_bias_act|{tuple(x.shape)}", 2.0 * x.element_size() * x.numel()):
            (_L().noise_bias_act_bf16 if x.dtype == BF16 else _L().noise_bias_act_f32)(
                _p(x), _p(bias), _p(noise), _p(nw), _p(y), x.numel() // c, c, alpha, scale, _st())
        ctx.save_for_backward(y, noise)
        ctx.cfg = (alpha, scale, bias is not None, nw is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        y, noise = ctx.saved_tensors
        alpha, scale, has_b, has_nw = ctx.cfg
        g = g.contiguous()
        c = y.shape[-1]
        gx = torch.empty_like(y)
        gnw = _zeros(1, y.device, torch.float32) if (has_nw and noise is not None) else None
        if y.dtype == BF16:  # one pass: gx, the bias gradient and the noise-weight gradient
            gb = torch.empty(c, device=y.device, dtype=torch.float32) if (has_b and ctx.needs_input_grad[1]) else None
            ws = _parts_ws(y.device, 2048 * (c + 8))
            _L().noise_bias_act_bwd_bf16(_p(g), _p(y), _p(noise) if gnw is not None else None, _p(gx), _p(gnw), _p(gb), _p(ws), ws.numel(),
                                         y.numel() // c, c, alpha, scale, _st())
            return gx, gb, None, gnw, None, None
        _L().noise_bias_act_bwd_f32(_p(g), _p(y), _p(noise) if gnw is not None else None, _p(gx), _p(gnw), y.numel() // c, c, alpha, scale, _st())
        gb = None
        if has_b and ctx.needs_input_grad[1]:
            gb = _zeros(c, y.device, torch.float32)
            _L().bias_grad_f32(_p(gx), y.numel() // c, c, c, _p(gb), _st())
        return gx, gb, None, gnw, None, None


def noise_bias_act(x, bias, noise=None, nw=None, alpha=0.2, scale=2 ** 0.5):
    return _NoiseBiasAct.apply(x, bias, noise, nw, float(alpha), float(scale))


class _UpFirDnNHWC(torch.autograd.Function):
    """upfirdn2d on [N,H,W,C]; the gradient is the same op with the flipped kernel, up/down swapped and the g_pad of
    op/upfirdn2d.py:108-113."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad, separable=False):
        _chk(x, dtype=x.dtype)
        _chk(kernel)
        n, h, w, c = x.shape
        kh, kw = kernel.shape
        # rank-one 4 x 4 taps on bf16 maps (the decoder's Blur): the separable LDS kernel, forward and adjoint
        ctx.sep = bool(separable) and x.dtype == BF16 and up == 1 and down == 1 and kh == 4 and kw == 4 and c % 32 == 0
        oh = (h * up + pad[0] + pad[1] - kh) // down + 1
        ow = (w * up + pad[0] + pad[1] - kw) // down + 1
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=x.dtype)
        # bandwidth kernel: the profile record carries algorithmic BYTES (in + out), tag prefix "bytes:"
        with _prof(f"bytes:upfirdn2d|{n}x{h}x{w}x{c} up{up} down{down}", float(x.element_size()) * (x.numel() + y.numel())):
            if ctx.sep:
                _L().blur_act_bf16(_p(x), _p(kernel), _p(y), n, h, w, c, pad[0], pad[1], pad[0], pad[1], None, None, None, None, 1.0, 1.0, 1, _st())
            else:
                (_L().upfirdn2d_nhwc_bf16 if x.dtype == BF16 else _L().upfirdn2d_nhwc_f32)(_p(x), _p(kernel), _p(y), n, h, w, c, kh, kw, up, up, down, down, pad[0], pad[1], pad[0], pad[1], _st())
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, (n, h, w, c), (oh, ow))
        return y

    @staticmethod
    def backward(ctx, g):
        (kernel,) = ctx.saved_tensors
        up, down, pad, (n, h, w, c), (oh, ow) = ctx.cfg
        kh, kw = kernel.shape
        gk = torch.flip(kernel, [0, 1]).contiguous()
        gx0, gy0 = kw - pad[0] - 1, kh - pad[0] - 1
        gx1 = w * up - ow * down + pad[0] - up + 1
        gy1 = h * up - oh * down + pad[0] - up + 1
        gx = torch.empty((n, h, w, c), device=g.device, dtype=g.dtype)
        with _prof(f"bytes:upfirdn2d_bwd|{n}x{h}x{w}x{c} up{up} down{down}", float(g.element_size()) * (g.numel() + gx.numel())):
            if ctx.sep:
                _L().blur_act_bf16(_p(g.contiguous()), _p(gk), _p(gx), n, oh, ow, c, gx0, gx1, gy0, gy1, None, None, None, None, 1.0, 1.0, 1, _st())
            else:
                (_L().upfirdn2d_nhwc_bf16 if g.dtype == BF16 else _L().upfirdn2d_nhwc_f32)(_p(g.contiguous()), _p(gk), _p(gx), n, oh, ow, c, kh, kw, down, down, up, up, gx0, gx1, gy0, gy1, _st())
        return gx, None, None, None, None, None


def upfirdn2d_nhwc(x, kernel, up=1, down=1, pad=(0, 0), separable=False):
    """separable: the caller states that the taps are an outer product (make_kernel of a 1-D list)"""
    return _UpFirDnNHWC.apply(x, kernel.contiguous(), int(up), int(down), (int(pad[0]), int(pad[1])), bool(separable))


# ---------------------------------------------------------------------------------------------------
# pSp encoder pieces (modules/psp/encoders/helpers.py)
# ---------------------------------------------------------------------------------------------------
class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = eltwise(EW_SIGMOID, x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return eltwise(EW_SIGMOID_BWD, g.contiguous(), y)


def sigmoid(x):
    return _Sigmoid.apply(x)


class _PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a):
        _chk(x, dtype=x.dtype)
        _chk(a)
        c = x.shape[-1]
        y = torch.empty_like(x)
        (_L().prelu_bf16 if x.dtype == BF16 else _L().prelu_f32)(_p(x), _p(a), _p(y), x.numel() // c, c, _st())
        ctx.save_for_backward(x, a)
        return y

    @staticmethod
    def backward(ctx, g):
        x, a = ctx.saved_tensors
        c = x.shape[-1]
        if x.dtype == BF16:
            gx, ga = torch.empty_like(x), torch.empty_like(a)
            ws = _parts_ws(x.device, 512 * c)
            _L().prelu_bwd_bf16(_p(g.contiguous()), _p(x), _p(a), _p(gx), _p(ga), _p(ws), ws.numel(), x.numel() // c, c, _st())
            return gx, ga
        gx, ga = torch.empty_like(x), _zeros_like(a)
        ws = _parts_ws(x.device, 1024 * c)
        _L().prelu_bwd_f32(_p(g.contiguous()), _p(x), _p(a), _p(gx), _p(ga), _p(ws), ws.numel(), x.numel() // c, c, _st())
        return gx, ga


def prelu(x, a):
    """nn.PReLU(C) on an NHWC tensor"""
    return _PReLU.apply(x, a)


class _Subsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        _chk(x, dtype=x.dtype)
        n, h, w, c = x.shape
        y = torch.empty((n, (h - 1) // stride + 1, (w - 1) // stride + 1, c), device=x.device, dtype=x.dtype)
        (_L().subsample_bf16 if x.dtype == BF16 else _L().subsample_f32)(_p(x), _p(y), n, h, w, c, stride, 0, _st())
        ctx.cfg = (x.shape, stride)
        return y

    @staticmethod
    def backward(ctx, g):
        (n, h, w, c), stride = ctx.cfg
        gx = torch.empty((n, h, w, c), device=g.device, dtype=g.dtype)
        (_L().subsample_bf16 if g.dtype == BF16 else _L().subsample_f32)(_p(g.contiguous()), _p(gx), n, h, w, c, stride, 1, _st())
        return gx, None


def subsample(x, stride):
    """nn.MaxPool2d(1, stride)"""
    return x if stride == 1 else _Subsample.apply(x, stride)


class _BatchNormTrain(torch.autograd.Function):
    """BatchNorm2d (training statistics) on NHWC: statistics over (N, H, W) per channel = the instance-norm kernels with the
    batch folded into the pixel axis.  ``groups`` > 1: the batch consists of that many equal parts that were SEPARATE forward
    calls in the reference (the source and the reference image of GradualStyleEncoder.forward, psp_encoders.py:101-125) -- each part
    gets its own statistics, exactly as if the parts had been normalised one after the other.
    Returns (y, stats[G][C][2] = (mean, rstd), sums[G][C][2] = fp64 (sum, sum of squares))."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, groups=1, passthrough=False):
        """passthrough: x itself is returned as a fourth output for x's OTHER consumer (the identity shortcut of an IR block); its
        gradient then arrives here and joins gx inside the backward kernel instead of in a separate accumulation pass"""
        b16 = x.dtype == BF16  # bf16 activations (IR-SE50 body of the pSp encoder): fp32 statistics / parameters, bf16 in and out
        _chk(x, dtype=x.dtype)
        _chk(gamma, beta)
        lib = _L()
        c = x.shape[-1]
        if x.shape[0] % groups:
            raise FmiError("grouped BatchNorm: the batch does not divide into the groups")
        rows = x.numel() // c // groups
        sums, ws = _norm_ws(x.device, groups, c)
        stats = torch.empty((groups, c, 2), device=x.device, dtype=torch.float32)
        (lib.instnorm_stats_bf16 if b16 else lib.instnorm_stats_f32)(_p(x), C.c_void_p(sums.data_ptr()), _p(stats), groups, rows, c, eps,
                                                                      C.c_void_p(ws.data_ptr()), ws.numel(), _st())
        y = torch.empty_like(x)
        (lib.instnorm_apply_bf16 if b16 else lib.instnorm_apply_f32)(_p(x), _p(stats), _p(gamma), _p(beta), _p(y), groups, rows, c, 1.0, _st())
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.groups = groups
        ctx.mark_non_differentiable(stats, sums)
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" for the statistics output (one fill launch per BatchNorm and step)
        if passthrough:
            return y, stats, sums, x.view_as(x)
        return y, stats, sums

    @staticmethod
    def backward(ctx, g, _gstats, _gsums, gpass=None):
        lib = _L()
        x, stats, gamma, beta = ctx.saved_tensors
        if g is None:  # only the pass-through branch carried a gradient
            return gpass, None, None, None, None, None
        c = x.shape[-1]
        groups = ctx.groups
        rows = x.numel() // c // groups
        g = g.contiguous()
        red, ws = _norm_ws(x.device, groups, c)
        gx, dg, db = torch.empty_like(x), _zeros_like(gamma), _zeros_like(beta)
        if x.dtype == BF16:
            lib.instnorm_bwd_reduce_bf16(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), groups, rows, c, 1.0,
                                         C.c_void_p(ws.data_ptr()), ws.numel(), _st())
            lib.instnorm_bwd_apply_bf16(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()),
                                        _p(gpass.contiguous()) if gpass is not None else None, _p(gx), _p(dg), _p(db), groups, rows, c, 1.0, _st())
            return gx, dg, db, None, None, None
        lib.instnorm_bwd_reduce_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), groups, rows, c, 1.0,
                                    C.c_void_p(ws.data_ptr()), ws.numel(), _st())
        if gpass is not None:
            lib.instnorm_bwd_apply_add_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), _p(gpass.contiguous()),
                                           _p(gx), _p(dg), _p(db), groups, rows, c, 1.0, _st())
        else:
            lib.instnorm_bwd_apply_f32(_p(x), _p(g), _p(stats), _p(gamma), _p(beta), C.c_void_p(red.data_ptr()), _p(gx), _p(dg), _p(db),
                                       groups, rows, c, 1.0, _st())
        return gx, dg, db, None, None, None


def batch_norm_running_update(stats, running_mean, running_var, num_batches_tracked, count, eps, momentum, sums=None):
    """in-place momentum update of the BatchNorm2d buffers from stats [1, C, 2] = (mean, rstd) of the batch (one launch); sums = the
    fp64 (sum, sum of squares) of the same pass, from which the variance is taken when given"""
    _chk(stats, running_mean, running_var)
    _chk(sums, dtype=torch.float64)
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise FmiError("num_batches_tracked must be an int64 device tensor")
    nbt = C.c_void_p(num_batches_tracked.data_ptr()) if num_batches_tracked is not None else None
    _L().batchnorm_running_update_f32(_p(stats), C.c_void_p(sums.data_ptr()) if sums is not None else None, _p(running_mean),
                                      _p(running_var), nbt, running_mean.numel(), int(count), float(eps), float(momentum), _st())


def batch_norm_train(x, gamma, beta, eps=1e-5, groups=1, passthrough=False):
    return _BatchNormTrain.apply(x, gamma, beta, float(eps), int(groups), bool(passthrough))


class _SplitBatch(torch.autograd.Function):
    """x[:n], x[n:] as views; the two gradients are joined by one copy"""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        return torch.cat([ga, gb], dim=0), None


def split_batch(x, n):
    return _SplitBatch.apply(x, int(n))


class _ChannelAffineFrozen(torch.autograd.Function):
    """y[..., c] = x * scale[c] + shift[c] with CONSTANT scale / shift (a frozen eval-mode BatchNorm) in one pass each way: the
    normalisation kernel with mean 0 / rstd 1 statistics forward, the channel scaling backward"""

    @staticmethod
    def forward(ctx, x, scale, shift, unit):
        _chk(x, scale, shift, unit)
        c = x.shape[-1]
        y = torch.empty_like(x)
        _L().instnorm_apply_f32(_p(x), _p(unit), _p(scale), _p(shift), _p(y), 1, x.numel() // c, c, 1.0, _st())
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = g.contiguous()
        c = g.shape[-1]
        gx = torch.empty_like(g)
        _L().scale_channels_f32(_p(g), _p(scale), _p(gx), 1, g.numel() // c, c, _st())
        return gx, None, None, None


class _ChannelAffineFrozenPass(torch.autograd.Function):
    """_ChannelAffineFrozen that also hands x on to its OTHER consumer (the identity shortcut of an IR block): (y, x'); the gradient
    coming back through x' is added inside the backward pass, gx = g * scale + g_x' (fmi_scale_channels_add_f32), instead of by an
    accumulation pass of the autograd engine"""

    @staticmethod
    def forward(ctx, x, scale, shift, unit):
        _chk(x, scale, shift, unit)
        c = x.shape[-1]
        y = torch.empty_like(x)
        _L().instnorm_apply_f32(_p(x), _p(unit), _p(scale), _p(shift), _p(y), 1, x.numel() // c, c, 1.0, _st())
        ctx.save_for_backward(scale)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g, gpass):
        (scale,) = ctx.saved_tensors
        g = g.contiguous()
        c = g.shape[-1]
        gx = torch.empty_like(g)
        if gpass is None:
            _L().scale_channels_f32(_p(g), _p(scale), _p(gx), 1, g.numel() // c, c, _st())
        else:
            _L().scale_channels_add_f32(_p(g), _p(scale), _p(gpass.contiguous()), _p(gx), 1, g.numel() // c, c, _st())
        return gx, None, None, None


_UNIT_STATS = {}


def channel_affine(x, scale, shift, passthrough=False):
    """y[..., c] = x * scale[c] + shift[c]  (BatchNorm2d in eval mode); gradients reach scale / shift when they require them.
    passthrough (constant scale / shift only): returns (y, x') -- see _ChannelAffineFrozenPass; None where that form does not apply"""
    n = x.shape[0]
    c = x.shape[-1]
    if x.dtype == torch.float32 and not scale.requires_grad and not shift.requires_grad and x.is_cuda:
        unit = _UNIT_STATS.get((x.device, c))
        if unit is None:  # stats[1][C][2] = (mean 0, rstd 1)
            unit = torch.tensor([0.0, 1.0], device=x.device).repeat(c).view(1, c, 2).contiguous()
            _UNIT_STATS[(x.device, c)] = unit
        if passthrough:
            return _ChannelAffineFrozenPass.apply(x.contiguous(), scale.contiguous(), shift.contiguous(), unit)
        return _ChannelAffineFrozen.apply(x.contiguous(), scale.contiguous(), shift.contiguous(), unit)
    if passthrough:
        return None
    y = scale_channels(x.reshape(1, -1, c), scale.view(1, c).contiguous())
    return _BiasAdd.apply(y.view(x.shape), shift.contiguous())


class _BiasAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b):
        c = x.shape[-1]
        y = torch.empty_like(x)
        _L().noise_bias_act_f32(_p(x), _p(b), None, None, _p(y), x.numel() // c, c, 1.0, 1.0, _st())
        return y

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.needs_input_grad[1]:
            g = g.contiguous()
            c = g.shape[-1]
            gb = _zeros(c, g.device, torch.float32)
            _L().bias_grad_f32(_p(g), g.numel() // c, c, c, _p(gb), _st())
        return g, gb


# ---------------------------------------------------------------------------------------------------
# multi-tensor Adam
# ---------------------------------------------------------------------------------------------------
def adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1, beta2, eps, weight_decay=0.0, guard=None):
    """step: python int (bias corrections computed on the host), or a 1-element int32 DEVICE tensor holding the number of steps taken
    so far (incremented by the launch; graph-capturable).  guard (device-counter form only): a device scalar -- the step becomes a
    no-op when it is not finite (train_psp.py:328-331)"""
    n = len(params)
    if n == 0:
        return
    if torch.is_tensor(step):
        if step.dtype != torch.int32 or not step.is_cuda or step.numel() != 1:
            raise FmiError("the device step counter must be a 1-element int32 device tensor")
        entries = (_lib.AdamEntry * n)()
        for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
            _chk(p, g, m, v)
            e = entries[i]
            e.p, e.g, e.m, e.v, e.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        if guard is not None:
            if guard.dtype != torch.float32 or not guard.is_cuda or guard.numel() != 1:
                raise FmiError("the guard must be a 1-element float32 device tensor")
            _L().adam_step_dev_guarded_f32(entries, n, lr, beta1, beta2, eps, weight_decay, C.c_void_p(step.data_ptr()), _p(guard), _st())
        else:
            _L().adam_step_dev_f32(entries, n, lr, beta1, beta2, eps, weight_decay, C.c_void_p(step.data_ptr()), _st())
        return
    if guard is not None:
        raise FmiError("a guarded step needs the device step counter (FusedAdam(capturable=True))")
    entries = (_lib.AdamEntry * n)()
    mx = 0
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        _chk(p, g, m, v)
        e = entries[i]
        e.p, e.g, e.m, e.v, e.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        mx = max(mx, p.numel())
    _L().adam_step_f32(entries, n, mx, lr, beta1, beta2, eps, weight_decay, step, _st())
