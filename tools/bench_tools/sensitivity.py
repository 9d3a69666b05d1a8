"""conditioning of the tiny-config loss gradients: d loss / d gen at gen0 and at gen0 * (1 + 1e-7 * noise), ONE library build (argv[1])"""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
_lib._LIB = _lib.Library(sys.argv[1])
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
G, D, gopt, optG, optD = T._tiny_models(fx, dev)
m = FF.binarise_mask(s["mask"].to(dev))
with torch.no_grad():
    gen0 = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
src, gt, ref = s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev)
d_sd = copy.deepcopy(D.state_dict())
for p in D.parameters():
    p.requires_grad_(False)
terms = {
    "perceptual": lambda g: gopt.vgg_loss(g, gt, lossType="perceptual"),
    "style": lambda g: gopt.style_loss(g, src, m),
    "contextual": lambda g: gopt.contextual_loss(g, ref, m),
    "gan": lambda g: gopt.gan_loss(D(g), True, False),
}
noise = torch.randn(gen0.shape, generator=torch.Generator().manual_seed(3)).to(dev)
for eps in (1e-7, 1e-6):
    for name, fn in terms.items():
        res = []
        for gin in (gen0, gen0 * (1 + eps * noise)):
            D.load_state_dict(d_sd)
            g = gin.clone().requires_grad_(True)
            loss = fn(g)
            (gr,) = torch.autograd.grad(loss, g)
            res.append((float(loss), gr))
        (la, ga), (lb, gb) = res
        print("eps %.0e %-11s loss rel diff %.2e   grad rel max diff %.2e  rel l2 %.2e" % (eps, name, abs(la - lb) / abs(lb), float((ga - gb).abs().max() / gb.abs().max()), float((ga - gb).norm() / gb.norm())))
