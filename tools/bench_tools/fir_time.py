"""channels-last Blur (upfirdn2d up = down = 1, 4x4) GB/s, fp32 and bf16, decoder shapes"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
k = torch.tensor([1.0, 3.0, 3.0, 1.0])
k = (k[None, :] * k[:, None] / 64 * 4).to(dev)
for dt in ((torch.bfloat16,) if 'bf16' in sys.argv else (torch.float32, torch.bfloat16)):
    for (n, h, c) in [(16, 65, 512), (16, 129, 256), (16, 257, 128), (4, 513, 64), (4, 1025, 32)]:
        x = torch.randn(n, h, h, c, device=dev).to(dt)
        for _ in range(3):
            y = FF.upfirdn2d_nhwc(x, k, pad=(1, 1))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(10):
            y = FF.upfirdn2d_nhwc(x, k, pad=(1, 1))
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        by = (x.numel() + y.numel()) * x.element_size()
        print(f"{str(dt)[6:]:9s} {n}x{h}x{h}x{c}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s", flush=True)
