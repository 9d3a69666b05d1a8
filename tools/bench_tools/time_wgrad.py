"""TFLOP/s of the convolution weight gradient (fmi_conv2d_wgrad_f32) at a few C2 shapes for each library build given on the command line"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(8, 128, 256, 128, 3), (8, 64, 128, 128, 3), (8, 256, 128, 64, 3), (8, 512, 64, 32, 3), (8, 32, 128, 128, 3), (8, 64, 256, 256, 3)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cases = []
for (n, h, ci, co, k) in SHAPES:
    x = torch.randn(n, h, h, ci, device=dev); gy = torch.randn(n, h, h, co, device=dev)
    d, oh, ow = FF.conv_desc(n, h, h, ci, co, k, k, 1, k // 2)
    dw = torch.zeros(k * k, ci, co, device=dev)
    cases.append((d, x, gy, dw, 2.0 * n * oh * ow * ci * co * k * k))
res = {}
for r in range(2):
    for path in sys.argv[1:]:
        lib = _lib.Library(path)
        out = []
        for d, x, gy, dw, fl in cases:
            t = timeit(lambda: lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(dw), None, 1, 0, st))
            out.append(fl / t / 1e9)
        res.setdefault(path, []).append(out)
print("%-24s " % "variant" + " ".join("%16s" % ("%dx%d^2 %d>%d" % (s[0], s[1], s[2], s[3])) for s in SHAPES))
for path, rs in res.items():
    best = [max(r[i] for r in rs) for i in range(len(rs[0]))]
    print("%-24s " % os.path.basename(path) + " ".join("%16.1f" % v for v in best))
