import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from face_mask_inpaint_amd import functional as FF
from test_gpu_model import _tiny_models
dev = torch.device("cuda:0")
fx = torch.load("/root/repo/tests/golden/picnet_train_tiny.pt", weights_only=True)
def run():
    G, D, gopt, optG, optD = _tiny_models(fx, dev)
    s = fx["step0"]
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    losses = gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)
    return gen.detach().double().cpu(), [float(l) for l in losses]
with FF.deterministic():
    gd, ld = run()
for i in range(3):
    g, l = run()
    print("gen max diff %.2e" % float((g - gd).abs().max()), ["%.2e" % abs(a / b - 1) for a, b in zip(l, ld)], ["%.3e" % v for v in l])
