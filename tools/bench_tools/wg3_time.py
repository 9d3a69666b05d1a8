"""weight gradient with both operands as piece images (wgrad_p3_kernel): TFLOP/s per layer shape of the C2 step.  FMI_WG3_TILE=1 forces the 8-wave tile."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
lib = _lib.lib()
#          n    h   ci   co  k
SHAPES = [(8, 512, 64, 32, 3), (8, 256, 128, 64, 3), (8, 64, 128, 128, 3), (8, 32, 128, 128, 3), (8, 256, 32, 32, 3), (8, 128, 256, 128, 3), (8, 128, 64, 128, 3),
          (8, 64, 256, 256, 3), (8, 32, 256, 256, 3), (8, 128, 32, 64, 3)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


tot = 0.0
for (n, h, ci, co, k) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(n, h, h, ci, device=dev); gy = torch.randn(n, h, h, co, device=dev)
    d, oh, ow = FF.conv_desc(n, h, h, ci, co, k, k, 1, k // 2)
    dw0 = torch.zeros(k * k, ci, co, device=dev); dw1 = torch.zeros_like(dw0)
    db = torch.zeros(co, device=dev)
    lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(dw0), FF._p(db), 1, 0, st)
    d.x3, d.y3 = FF.p3_of(x).data_ptr(), FF.p3_of(gy).data_ptr()
    lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(dw1), None, 1, 0, st)
    diff = (dw0 - dw1).abs().max().item() / dw0.abs().max().item()
    t = timeit(lambda: lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(dw1), None, 1, 1, st))
    fl = 2.0 * n * oh * ow * ci * co * k * k
    tot += t
    print("%-24s %8.1f TFLOP/s %8.1f us  maxdiff %.1e" % ("%dx%d^2 %d>%d k%d" % (n, h, ci, co, k), fl / t / 1e9, t * 1e3, diff))
print("total %.3f ms" % tot)
