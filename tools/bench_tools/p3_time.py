"""convolution forward with the activation operand as a bf16 piece image (fmi_conv_desc.x3) against the in-wave split:
max difference of the results, TFLOP/s of both, and the cost of the standalone split pass.  FMI_P3_TILE=1/2/3 forces a tile."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
lib = _lib.lib()
#          n    h   ci   co  k  stride
SHAPES = [(8, 128, 256, 256, 3, 1), (24, 56, 256, 256, 3, 1), (24, 28, 512, 512, 3, 1), (24, 224, 64, 64, 3, 1), (24, 112, 128, 128, 3, 1),
          (8, 32, 128, 128, 3, 1), (8, 64, 128, 128, 3, 1), (8, 256, 32, 32, 3, 1), (8, 512, 64, 32, 3, 1), (8, 128, 256, 64, 1, 1)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def pieces_w(w, k, ci, co):  # [3][taps][C/8][K][8]
    x0 = w.bfloat16(); r1 = w - x0.float(); x1 = r1.bfloat16(); x2 = (r1 - x1.float()).bfloat16()
    return torch.stack([t.view(k * k, ci // 8, 8, co).permute(0, 1, 3, 2).contiguous() for t in (x0, x1, x2)])


print("%-28s %10s %10s %10s %10s %10s %10s" % ("shape", "base TF", "p3 TF", "p3+y3 TF", "split us", "maxdiff", "y3 ok"))
for (n, h, ci, co, k, s) in SHAPES[:int(os.environ.get("NSHAPES", "99"))]:
    torch.manual_seed(0)
    x = torch.randn(n, h, h, ci, device=dev); w = torch.randn(k * k, ci, co, device=dev) * 0.05
    w3 = pieces_w(w, k, ci, co)
    d, oh, ow = FF.conv_desc(n, h, h, ci, co, k, k, s, k // 2)
    d.w3 = w3.data_ptr()
    y0 = torch.empty(n, oh, ow, co, device=dev); y1 = torch.empty_like(y0)
    x3 = torch.empty(x.numel() * 3, device=dev, dtype=torch.bfloat16)
    y3 = torch.empty(y0.numel() * 3, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * n * oh * ow * ci * co * k * k
    t_split = timeit(lambda: lib.split3_f32(FF._p(x), C.c_void_p(x3.data_ptr()), None, n * h * h, ci, 0, 0.0, st))
    t0 = timeit(lambda: lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y0), 0, 1, 0, st))
    d.x3 = x3.data_ptr()
    t1 = timeit(lambda: lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y1), 0, 1, 0, st))
    diff = (y0 - y1).abs().max().item() / y0.abs().max().item()
    t2, ok = float("nan"), "-"
    if co % 16 == 0:
        d.y3 = y3.data_ptr()
        t2 = timeit(lambda: lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y1), 0, 1, 0, st))
        back = torch.empty_like(y1)
        lib.merge3_f32(C.c_void_p(y3.data_ptr()), FF._p(back), n * oh * ow, co, st)
        ok = str(bool(torch.equal(back, y1)))
    print("%-28s %10.1f %10.1f %10.1f %10.1f %10.2e %10s" % ("%dx%d^2 %d>%d k%d" % (n, h, ci, co, k), fl / t0 / 1e9, fl / t1 / 1e9, fl / t2 / 1e9, t_split * 1e3, diff, ok))
