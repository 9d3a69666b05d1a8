"""dense batched GEMM: candidate library (argv[1]) vs reference build (argv[2]) vs torch, over layouts / tails / batch / split-K"""
import ctypes as C, os, sys, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
A, B = _lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])
g = torch.Generator().manual_seed(0)
bad = 0
cases = [(128, 128, 64, 1), (100, 36, 52, 3), (64, 256, 784, 8), (784, 784, 64, 8), (256, 256, 3136, 4), (64, 64, 50176, 2), (512, 512, 784, 2),
         (36, 100, 20, 2), (8, 512, 512, 1), (512, 8, 512, 1), (32, 32, 16384, 1), (16384, 64, 32, 2), (132, 260, 68, 2)]
for (M, N, K, bt), ta, tb in itertools.product(cases, (0, 1), (0, 1)):
    a = torch.randn(bt, K, M, generator=g).to(dev) if ta else torch.randn(bt, M, K, generator=g).to(dev)
    b = torch.randn(bt, N, K, generator=g).to(dev) if tb else torch.randn(bt, K, N, generator=g).to(dev)
    am = a.transpose(1, 2) if ta else a
    bm = b.transpose(1, 2) if tb else b
    c0 = torch.randn(bt, M, N, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    want = 0.5 * (am.double() @ bm.double()) + 0.25 * c0.double() + bias.double()
    outs = []
    for lib in (A, B):
        c = c0.clone()
        sa = (1, M) if ta else (K, 1)
        sb = (1, K) if tb else (N, 1)
        lib.gemm_f32(FF._p(a), FF._p(b), FF._p(c), M, N, K, sa[0], sa[1], sb[0], sb[1], N, 1, bt, M * K, K * N, M * N, 0.5, 0.25, FF._p(bias), st)
        torch.cuda.synchronize()
        outs.append(c)
    ea = float((outs[0].double() - want).abs().max() / want.abs().max())
    eb = float((outs[1].double() - want).abs().max() / want.abs().max())
    flag = "  <-- BAD" if ea > 1e-5 else ""
    bad += ea > 1e-5
    print((M, N, K, bt), "ta", ta, "tb", tb, "cand %.1e ref %.1e%s" % (ea, eb, flag))
print("BAD" if bad else "OK", bad)
