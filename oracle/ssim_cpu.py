"""ORACLE (test infrastructure): CPU restatement of modules/evaluations/ssim.py:8-38 (11-tap gaussian sigma 1.5,
zero-padded depth-wise filtering, C1 = 0.01^2, C2 = 0.03^2).  Pinned by tests/golden/ssim.pt (reference's own ssim)."""
import torch
import torch.nn.functional as F


def ssim(img1, img2, window_size=11, size_average=True):
    c = img1.shape[1]
    xs = torch.arange(window_size, dtype=torch.float64) - window_size // 2
    g = torch.exp(-xs ** 2 / (2 * 1.5 ** 2))
    g = (g / g.sum()).float()
    win = (g[:, None] * g[None, :]).expand(c, 1, window_size, window_size).contiguous()
    p = window_size // 2
    f = lambda t: F.conv2d(t, win, padding=p, groups=c)
    mu1, mu2 = f(img1), f(img2)
    s11, s22, s12 = f(img1 * img1) - mu1 * mu1, f(img2 * img2) - mu2 * mu2, f(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)
