"""full tiny training step 0 under two library builds: G gradient differences; then forward_multi terms"""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
libs = [_lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])]
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
res = []
for lib in libs:
    _lib._LIB = lib
    G, D, gopt, optG, optD = T._tiny_models(fx, dev)
    gopt.lambda_cx = float(os.environ.get("LCX", "1"))
    gopt.lambda_perc = float(os.environ.get("LPERC", "0.1"))
    gopt.lambda_style = float(os.environ.get("LSTY", "250"))
    gopt.lambda_g = float(os.environ.get("LG", "0.01"))
    cap = {}
    orig = optG.step
    def step(closure=None, orig=orig, G=G, cap=cap):
        cap.update({n: p.grad.detach().clone() for n, p in G.named_parameters() if p.grad is not None})
        return orig(closure)
    optG.step = step
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    out = gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)
    res.append((cap, [float(v) for v in out], gen.detach().clone()))
print("losses", res[0][1], res[1][1])
rows = sorted(((float((res[0][0][n] - v).abs().max() / (v.abs().max() + 1e-30)), float(v.abs().max()), n) for n, v in res[1][0].items()), reverse=True)
for e, mx, n in rows[:14]:
    print("  %.2e  max|g| %.2e %s" % (e, mx, n))
