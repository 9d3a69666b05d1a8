"""times the dense GEMM and a few conv shapes for each library variant given on the command line (paths)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(8, 128, 256, 256, 3), (24, 56, 256, 256, 3), (8, 256, 32, 32, 3), (8, 32, 128, 128, 3), (24, 224, 64, 64, 3)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev); c = torch.empty(4096, 4096, device=dev)
convs = []
for (n, h, ci, co, k) in SHAPES:
    x = torch.randn(n, h, h, ci, device=dev); w = torch.randn(k * k, ci, co, device=dev) * 0.05
    d, oh, ow = FF.conv_desc(n, h, h, ci, co, k, k, 1, k // 2)
    y = torch.empty(n, oh, ow, co, device=dev)
    convs.append((d, x, w, y, 2.0 * n * oh * ow * ci * co * k * k))
    if os.environ.get("W3"):  # the bf16 piece images of the weights: [3][taps][C/8][K][8]
        x0 = w.bfloat16(); r1 = w - x0.float(); x1 = r1.bfloat16(); x2 = (r1 - x1.float()).bfloat16()
        w3 = torch.stack([t.view(k * k, ci // 8, 8, co).permute(0, 1, 3, 2).contiguous() for t in (x0, x1, x2)])
        d.w3 = w3.data_ptr()
        keep = globals().setdefault("_keep", [])
        keep.append(w3)
a2 = torch.randn(131072, 2304, device=dev); b2 = torch.randn(2304, 256, device=dev); c2 = torch.empty(131072, 256, device=dev)
rounds = int(os.environ.get("ROUNDS", "2"))
res = {}
for r in range(rounds):
    for path in sys.argv[1:]:
        lib = _lib.Library(path)
        out = []
        t = timeit(lambda: lib.gemm_f32(FF._p(a), FF._p(b), FF._p(c), 4096, 4096, 4096, 4096, 1, 4096, 1, 4096, 1, 1, 0, 0, 0, 1.0, 0.0, None, st))
        out.append(2 * 4096 ** 3 / t / 1e9)
        t = timeit(lambda: lib.gemm_f32(FF._p(a2), FF._p(b2), FF._p(c2), 131072, 256, 2304, 2304, 1, 256, 1, 256, 1, 1, 0, 0, 0, 1.0, 0.0, None, st))
        out.append(2 * 131072 * 256 * 2304 / t / 1e9)
        for d, x, w, y, fl in convs:
            t = timeit(lambda: lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y), 0, 1, 0, st))
            out.append(fl / t / 1e9)
        res.setdefault(path, []).append(out)
print("%-28s %8s %14s " % ("variant", "gemm4k", "gemm131kx256x2304") + " ".join("%14s" % ("%dx%d^2 %d>%d" % (s[0], s[1], s[2], s[3])) for s in SHAPES))
for path, rs in res.items():
    best = [max(r[i] for r in rs) for i in range(len(rs[0]))]
    print("%-28s " % os.path.basename(path) + " ".join("%8.1f" % v if i == 0 else "%14.1f" % v for i, v in enumerate(best)))
