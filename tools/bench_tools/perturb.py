"""ONE build (argv[1]): G parameter gradients (L1 loss only) with and without a 1e-7 relative perturbation of one early conv output"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
_lib._LIB = _lib.Library(sys.argv[1])
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
orig_conv = FF.conv2d
res = []
for eps in (0.0, 1e-7, 1e-7, 3e-7):
    cnt = [0]
    gen_ = torch.Generator().manual_seed(len(res))
    def conv(*a, **k):
        out = orig_conv(*a, **k)
        cnt[0] += 1
        if cnt[0] == 7 and eps:  # the 1x1 16->16 convolution at 32x32 (GEMM launch 6)
            out = out * (1 + eps * torch.randn(out.shape, generator=gen_).to(out.device))
        return out
    FF.conv2d = conv
    G, D, gopt, optG, optD = T._tiny_models(fx, dev)
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    FF.l1_loss(FF.to_nhwc(gen), FF.to_nhwc(s["gt"].to(dev))).backward()
    torch.cuda.synchronize()
    res.append({n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None and p.ndim > 1})
for i in range(1, len(res)):
    rows = sorted(((float((res[i][n] - v).abs().max() / (v.abs().max() + 1e-30)), float((res[i][n] - v).norm() / v.norm()), n) for n, v in res[0].items()), reverse=True)
    print("perturbation run %d: worst max-rel %.2e (L2 %.2e) %s ; median max-rel %.2e" % (i, rows[0][0], rows[0][1], rows[0][2], rows[len(rows) // 2][0]))
