"""the reference's native upfirdn2d through its drop-in name, planar layout, Blur shapes; GB/s of in + out"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd.modules.psp.stylegan2.op.upfirdn2d import _native
dev = torch.device("cuda:0")
k = torch.tensor([1.0, 3.0, 3.0, 1.0])
k = (k[None, :] * k[:, None] / 64 * 4).to(dev)
for dt in (torch.float32, torch.bfloat16):
    for (m, h) in [(128, 1025), (256, 513), (2048, 129), (8192, 65)]:
        x = torch.randn(m, h, h, device=dev).to(dt)
        for _ in range(3):
            y = _native(x, k, 1, 1, 1, 1, 1, 1, 1, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(10):
            y = _native(x, k, 1, 1, 1, 1, 1, 1, 1, 1)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        by = (x.numel() + y.numel()) * x.element_size()
        print(f"{str(dt)[6:]:9s} {m}x{h}x{h} -> {tuple(y.shape)}: {ms:.3f} ms  {by / ms / 1e6:.0f} GB/s", flush=True)
