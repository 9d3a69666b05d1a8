"""step-0 gradient error of the tiny golden config, per parameter (relative to the tensor's largest entry)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
for rep in range(int(os.environ.get("REPS", "2"))):
    G, D, gopt, optG, optD = T._tiny_models(fx, dev)
    cap = {}
    orig = optG.step
    def step(closure=None, orig=orig):
        cap.update({n: p.grad.detach().cpu().clone() for n, p in G.named_parameters() if p.grad is not None})
        return orig(closure)
    optG.step = step
    s = fx["step0"]
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)
    print("gen err", float((gen.detach().cpu() - s["gen"]).abs().max()))
    rows = []
    for n, g in s["G_grads"].items():
        rows.append((float((cap[n] - g).abs().max() / (g.abs().max() + 1e-30)), float(g.abs().max()), n))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print("%.3e  max|g| %.3e  %s" % r)
    print()
