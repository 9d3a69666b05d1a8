"""compares conv fwd / dgrad / wgrad of two library variants (argv[1] = candidate, argv[2] = reference build) on many shapes"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
A, B = _lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])
g = torch.Generator(device="cpu").manual_seed(0)
bad = 0
for (n, h, w, c, k, ks, stride, pad) in [(2, 32, 32, 16, 16, 1, 1, 0), (2, 16, 16, 16, 16, 1, 1, 0), (2, 32, 32, 16, 32, 1, 1, 0), (2, 32, 32, 16, 64, 1, 1, 0), (2, 32, 32, 16, 128, 1, 1, 0), (8, 32, 32, 16, 16, 1, 1, 0), (2, 64, 64, 16, 16, 1, 1, 0),(2, 64, 64, 8, 8, 3, 1, 1), (2, 64, 64, 8, 16, 3, 1, 1), (2, 32, 32, 16, 16, 3, 1, 1), (2, 16, 16, 32, 32, 3, 1, 1),
                                         (2, 33, 31, 16, 48, 3, 1, 1), (2, 32, 32, 32, 16, 1, 1, 0), (3, 20, 20, 64, 36, 3, 2, 1), (2, 8, 8, 32, 32, 3, 1, 1),
                                         (1, 40, 40, 16, 4, 3, 1, 1), (2, 24, 24, 12, 20, 3, 1, 1), (2, 64, 64, 32, 8, 3, 1, 1), (8, 32, 32, 128, 128, 3, 1, 1),
                                         (2, 17, 19, 48, 64, 5, 1, 2), (4, 128, 128, 32, 32, 3, 1, 1),
                                         (2, 32, 32, 16, 32, 3, 2, 1), (2, 32, 32, 32, 64, 3, 2, 1), (2, 32, 32, 16, 16, 4, 2, 1), (2, 31, 33, 16, 48, 3, 2, 1), (2, 16, 16, 256, 256, 3, 2, 1)]:
    x = torch.randn(n, h, w, c, generator=g).to(dev); wf = (torch.randn(ks * ks, c, k, generator=g) * 0.1).to(dev)
    wt = wf.permute(0, 2, 1).contiguous()
    d, oh, ow = FF.conv_desc(n, h, w, c, k, ks, ks, stride, pad)
    gy = torch.randn(n, oh, ow, k, generator=g).to(dev)
    outs = []
    for lib in (A, B):
        y = torch.empty(n, oh, ow, k, device=dev)
        lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(wf), None, None, FF._p(y), 0, 1, 0, st)
        gx = torch.empty_like(x)
        lib.conv2d_dgrad_f32(C.byref(d), FF._p(gy), FF._p(wt), None, None, FF._p(gx), 1, 0, st)
        gw = torch.zeros_like(wf); gb = torch.zeros(k, device=dev)
        fuse = (ks * ks * c) % 4 == 0
        lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(gw), FF._p(gb) if fuse else None, 1, 0, st)
        torch.cuda.synchronize()
        outs.append((y, gx, gw, gb))
    msg = []
    for name, a, b in zip(("fwd", "dgrad", "wgrad", "dbias"), outs[0], outs[1]):
        e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        msg.append("%s %.1e" % (name, e))
        bad += e > 1e-4
    print((n, h, w, c, k, ks, stride, pad), "  ".join(msg))
print("BAD" if bad else "OK", bad)
