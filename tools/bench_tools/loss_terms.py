"""d(loss term)/d(gen) of the tiny golden config under two library builds (argv[1] candidate, argv[2] reference)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
libs = [_lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])]
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
_lib._LIB = libs[1]
G, D, gopt, optG, optD = T._tiny_models(fx, dev)
m = FF.binarise_mask(s["mask"].to(dev))
with torch.no_grad():
    gen0 = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
src, gt, ref = s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev)
terms = {
    "perceptual": lambda g: gopt.vgg_loss(g, gt, lossType="perceptual"),
    "style": lambda g: gopt.style_loss(g, src, m),
    "contextual": lambda g: gopt.contextual_loss(g, ref, m),
    "l1": lambda g: FF.l1_loss(FF.to_nhwc(g), FF.to_nhwc(gt)),
    "gan": lambda g: gopt.gan_loss(D(g), True, False),
}
import copy
d_sd = copy.deepcopy(D.state_dict())
for name, fn in terms.items():
    res = []
    for lib in libs:
        _lib._LIB = lib
        D.load_state_dict(d_sd)
        gopt.vgg_loss._cache = {}
        g = gen0.clone().requires_grad_(True)
        loss = fn(g)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), g.grad.clone()))
    (la, ga), (lb, gb) = res
    print("%-11s loss %.6e vs %.6e   grad rel max err %.2e   rel l2 %.2e" % (name, la, lb, float((ga - gb).abs().max() / gb.abs().max()), float((ga - gb).norm() / gb.norm())))

# the generator's own backward: d(sum(gen * r)) / d(params) under both builds
r = torch.randn(gen0.shape, generator=torch.Generator().manual_seed(1)).to(dev)
g_sd = copy.deepcopy(G.state_dict())
res = []
for lib in libs:
    _lib._LIB = lib
    G.load_state_dict(g_sd)
    G.zero_grad()
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    (gen * r).sum().backward()
    torch.cuda.synchronize()
    res.append((gen.detach().clone(), {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}))
print("gen fwd rel err %.2e" % float((res[0][0] - res[1][0]).abs().max() / res[1][0].abs().max()))
rows = sorted(((float((res[0][1][n] - v).abs().max() / (v.abs().max() + 1e-30)), n) for n, v in res[1][1].items()), reverse=True)
for e, n in rows[:8]:
    print("  %.2e %s" % (e, n))
