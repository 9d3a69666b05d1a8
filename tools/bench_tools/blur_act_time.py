"""fmi_blur_act_bf16 at the decoder's Blur shapes: plain FIR / separable FIR / separable + demodulation + noise + bias + lrelu, GB/s of in + out"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import _lib, functional as FF
lib = _lib.lib()
dev = torch.device("cuda:0")
k = torch.tensor([1.0, 3.0, 3.0, 1.0])
k = (k[None, :] * k[:, None] / 64 * 4).to(dev)
for (n, h, c) in [(16, 65, 512), (16, 129, 256), (16, 257, 128), (4, 513, 64), (4, 1025, 32)]:
    x = torch.randn(n, h, h, c, device=dev).bfloat16()
    y = torch.empty(n, h - 1, h - 1, c, device=dev, dtype=torch.bfloat16)
    d, noise, nw, b = torch.rand(n, c, device=dev) + 0.5, torch.randn(n, h - 1, h - 1, device=dev), torch.randn(1, device=dev), torch.randn(c, device=dev)
    st = FF._st()
    res = []
    for args in ((None, None, None, None, 1.0, 1.0, 0), (None, None, None, None, 1.0, 1.0, 1), (d, noise, nw, b, 0.2, 2 ** 0.5, 1)):
        fn = lambda: lib.blur_act_bf16(FF._p(x), FF._p(k), FF._p(y), n, h, h, c, 1, 1, 1, 1, FF._p(args[0]), FF._p(args[1]), FF._p(args[2]), FF._p(args[3]), args[4], args[5], args[6], st)
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        res.append((x.numel() + y.numel()) * 2 / ms / 1e6)
    print(f"{n}x{h}x{h}x{c}: plain {res[0]:.0f}  separable {res[1]:.0f}  separable+output stage {res[2]:.0f} GB/s", flush=True)
