"""producer kernel -> 1x1 / 3x3 conv without host sync in between, candidate (argv[1]) vs reference build (argv[2])"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
A, B = _lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])
g = torch.Generator().manual_seed(0)
n, h, c, k = 2, 32, 16, 16
wf = (torch.randn(1, c, k, generator=g) * 0.3).to(dev); bias = torch.randn(k, generator=g).to(dev)
d, oh, ow = FF.conv_desc(n, h, h, c, k, 1, 1, 1, 0)
worst = 0.0
for it in range(200):
    x0 = torch.randn(n, h, h, c, generator=g).to(dev)
    outs = []
    for lib in (A, B):
        x = torch.empty_like(x0); y = torch.empty(n, oh, ow, k, device=dev)
        torch.cuda.synchronize()
        lib.eltwise_f32(0, FF._p(x0), FF._p(x0), FF._p(x), x0.numel(), 0.0, st) if hasattr(lib, "eltwise_f32") else x.copy_(x0 * 2)
        lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(wf), FF._p(bias), None, FF._p(y), 0, 1, 0, st)
        torch.cuda.synchronize()
        outs.append((x.clone(), y))
    assert torch.equal(outs[0][0], outs[1][0])
    e = float((outs[0][1] - outs[1][1]).abs().max() / outs[1][1].abs().max())
    worst = max(worst, e)
print("worst rel err over 200 producer->conv sequences: %.2e" % worst)
