import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from face_mask_inpaint_amd import functional as FF
import torch.nn.functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (n, c, k, h, ks, st, pad) in ((4, 64, 128, 32, 1, 2, 0), (4, 128, 256, 16, 1, 2, 0), (4, 256, 512, 8, 1, 2, 0), (4, 64, 64, 32, 3, 2, 1), (4, 64, 64, 33, 3, 2, 1),
                                  (32, 64, 64, 128, 3, 1, 1), (32, 512, 512, 16, 3, 1, 1), (4, 64, 128, 31, 1, 2, 0)):
    x = torch.randn(n, c, h, h, generator=g).to(torch.bfloat16)
    w = (torch.randn(k, c, ks, ks, generator=g) / (c * ks * ks) ** 0.5)
    xr = x.float().requires_grad_(True); wr = w.to(torch.bfloat16).float().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=st, padding=pad)
    gy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
    y.backward(gy.float())
    # poison the allocator so that unwritten outputs show
    junk = torch.full((64 << 20,), float("nan"), device=dev); del junk
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    (pw,) = FF.prepare_weights([(wd, None, None)])
    yd = FF.conv2d(xd, pw, None, None, st, pad)
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    e_y = float((yd.detach().float().cpu().permute(0, 3, 1, 2) - y.detach()).abs().max() / y.abs().max())
    gx = xd.grad.float().cpu().permute(0, 3, 1, 2)
    e_gx = float((gx - xr.grad).abs().max() / xr.grad.abs().max())
    e_gw = float((wd.grad.cpu() - wr.grad).abs().max() / wr.grad.abs().max())
    print((n, c, k, h, ks, st, pad), "y %.2e gx %.2e gw %.2e finite %s" % (e_y, e_gx, e_gw, bool(torch.isfinite(gx).all())))
