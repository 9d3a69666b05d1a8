"""TEST INFRASTRUCTURE ONLY (oracle): SSIM / MS-SSIM as the trainers' metric package defines them, in plain torch on the CPU.

The reference imports ``SSIM`` / ``MS_SSIM`` from pytorch_msssim (train_reference_fill.py:17,207-209, train_psp.py:16,176-178,
PICNet_inference.py:8,130-131) -- a third-party package that is neither vendored in /root/reference nor installed here, and un-pinned
there (scripts/env_setup.sh:38 "pip install pytorch-msssim").  Restated from its published algorithm (Wang et al. 2003 multi-scale SSIM;
the package's conventions: separable 11-tap Gaussian of sigma 1.5 applied without padding, K = (0.01, 0.03), 2 x 2 mean pool with
padding = size % 2 between scales, relu on the cs / ssim terms, weights 0.0448 0.2856 0.3001 0.2363 0.1333).
**Parity unpinned**: the reference holds no golden value for either metric and the package cannot be run here."""
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _window(size=11, sigma=1.5):
    x = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return (g / g.sum())


def _filter(x, g):
    c = x.shape[1]
    x = F.conv2d(x, g.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.conv2d(x, g.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)


def _ssim_cs(x, y, g, c1, c2):
    mu1, mu2 = _filter(x, g), _filter(y, g)
    s11 = _filter(x * x, g) - mu1 * mu1
    s22 = _filter(y * y, g) - mu2 * mu2
    s12 = _filter(x * y, g) - mu1 * mu2
    cs = (2 * s12 + c2) / (s11 + s22 + c2)
    ss = (2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1) * cs
    return ss.flatten(2).mean(-1), cs.flatten(2).mean(-1)


def ssim(x, y, data_range=1.0, size_average=True):
    g = _window().to(x.dtype)
    s, _ = _ssim_cs(x, y, g, (0.01 * data_range) ** 2, (0.03 * data_range) ** 2)
    return s.mean() if size_average else s.mean(1)


def ms_ssim(x, y, data_range=1.0, size_average=True, weights=WEIGHTS):
    g = _window().to(x.dtype)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    terms = []
    for lv in range(len(weights)):
        s, cs = _ssim_cs(x, y, g, c1, c2)
        if lv < len(weights) - 1:
            terms.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
    terms.append(torch.relu(s))
    val = torch.prod(torch.stack(terms) ** torch.tensor(weights, dtype=x.dtype).view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)
