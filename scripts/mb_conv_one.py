import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
from face_mask_inpaint_amd import functional as FF, _lib
lib = _lib.lib(); dev = torch.device('cuda:0')
n, h, c, k = 8, 128, 256, 256
x = torch.randn(n, h, h, c, device=dev); w = torch.randn(9, c, k, device=dev) * 0.05
d, oh, ow = FF.conv_desc(n, h, h, c, k, 3, 3, 1, 1)
y = torch.empty(n, oh, ow, k, device=dev)
for _ in range(3):
    lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y), 0, 1, 0, FF._st())
torch.cuda.synchronize()
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev); cc = torch.empty(4096, 4096, device=dev)
for _ in range(3):
    FF.gemm_raw(FF._p(a), FF._p(b), FF._p(cc), 4096, 4096, 4096, (4096, 1), (4096, 1), (4096, 1))
torch.cuda.synchronize()
