"""one line: forward / adjoint TFLOP/s of the eight-phase bf16 convolution kernel (mode 8) on the decoder's three big layers"""
import ctypes as C, sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import _lib, functional as FF
lib = _lib.lib()
dev = torch.device("cuda:0")
lib.debug_bf16_tile(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
out = []
for (n, c, k, h) in [(16, 512, 512, 64), (16, 256, 256, 128), (16, 128, 128, 256)]:
    d, oh, ow = FF.conv_desc(n, h, h, c, k, 3, 3, 1, 1, 0)
    x = torch.randn(n, h, h, c, device=dev).bfloat16()
    gy = torch.randn(n, h, h, k, device=dev).bfloat16()
    wnk = (torch.randn(k, 9, c, device=dev) / 70).bfloat16()
    wck = (torch.randn(c, 9, k, device=dev) / 70).bfloat16()
    y, dx = torch.empty_like(gy), torch.empty_like(x)
    st = FF._st()
    for fn in (lambda: lib.conv2d_fwd_bf16(C.byref(d), FF._p(x), FF._p(wnk), None, FF._p(y), None, 0, st),
               lambda: lib.conv2d_dgrad_bf16(C.byref(d), FF._p(gy), FF._p(wck), None, FF._p(dx), None, 0, st)):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        out.append(2.0 * n * h * h * k * c * 9 / (s.elapsed_time(e) / 20) / 1e9)
print("  ".join(f"{t:.0f}" for t in out), flush=True)
