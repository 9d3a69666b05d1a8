import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
n, h, c, k = 8, 1024, 32, 3
x = torch.randn(n, h, h, c, device=dev); wf = torch.randn(9, c, k, device=dev) * 0.05; wt = wf.permute(0, 2, 1).contiguous()
d, oh, ow = FF.conv_desc(n, h, h, c, k, 3, 3, 1, 1, 1)
y = torch.empty(n, h, h, k, device=dev); gy = torch.randn(n, h, h, k, device=dev); gx = torch.empty_like(x)
gw = torch.zeros_like(wf); gb = torch.zeros(k, device=dev)
def timeit(fn, nrep=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(nrep): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / nrep
for path in sys.argv[1:]:
    lib = _lib.Library(path)
    tf = timeit(lambda: lib.conv2d_thin_fwd_f32(C.byref(d), FF._p(x), FF._p(wf), None, None, FF._p(y), 1, st))
    tf0 = timeit(lambda: lib.conv2d_thin_fwd_f32(C.byref(d), FF._p(x), FF._p(wf), None, None, FF._p(y), 0, st))
    print("   forward without tanh: %.3f ms" % tf0)
    td = timeit(lambda: lib.conv2d_thin_dgrad_f32(C.byref(d), FF._p(gy), FF._p(wt), FF._p(gx), st))
    tw = timeit(lambda: lib.conv2d_thin_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(gw), FF._p(gb), st))
    gb_ = x.numel() * 4 / 1e9
    print("%-40s fwd %.3f ms (%.2f TB/s)  dgrad %.3f ms (%.2f TB/s)  wgrad %.3f ms (%.2f TB/s)" % (os.path.basename(path), tf, gb_ / tf, td, gb_ / td, tw, gb_ / tw))
