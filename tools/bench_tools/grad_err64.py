"""per-parameter gradient error of the tiny golden training steps against the reference's FLOAT64 evaluation
(tests/golden/picnet_train_tiny.pt: G_grads64 / D_grads64), next to the reference's own fp32 error: step 0 from G_sd0 / D_sd0,
step 1 from the fp32 reference's state at the start of step 1 (G_sd1 / D_sd1)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
for step in (0, 1):
    f2 = dict(fx)
    f2["G_sd0"], f2["D_sd0"] = fx[f"G_sd{step}"], fx[f"D_sd{step}"]
    G, D, gopt, optG, optD = T._tiny_models(f2, dev)
    cap = {"G": {}, "D": {}}
    for key, opt, mod in (("G", optG, G), ("D", optD, D)):
        orig = opt.step
        def stepf(closure=None, orig=orig, key=key, mod=mod):
            cap[key].update({n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters() if p.grad is not None})
            return orig(closure)
        opt.step = stepf
    s = fx[f"step{step}"]
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    out = gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)
    print(f"step {step}: gen err vs fp64 {float((gen.detach().cpu() - s['gen64']).abs().max()):.3e} (reference fp32: {float((s['gen'] - s['gen64']).abs().max()):.3e})")
    print("  losses rel err vs fp64:", ["%.2e" % abs(float(o) / float(w) - 1) for o, w in zip(out, s["losses64"])],
          " reference fp32:", ["%.2e" % abs(float(s[k]) / float(w) - 1) for k, w in zip(("d_loss", "g_loss", "perc", "style", "cx"), s["losses64"])])
    for key in ("G", "D"):
        rows = []
        for n, g64 in s[f"{key}_grads64"].items():
            mx = float(g64.abs().max())
            if mx < 1e-12 or n not in cap[key]:
                continue
            eh = float((cap[key][n] - g64).abs().max()) / mx
            er = float((s[f"{key}_grads"][n] - g64).abs().max()) / mx
            rows.append((eh, er, eh / max(er, 1e-30), mx, n))
        rows.sort(reverse=True)
        print(f"  {key}: worst HIP err {rows[0][0]:.2e}, worst reference-fp32 err {max(r[1] for r in rows):.2e}, tensors with HIP err > 2 x own ref err: {sum(1 for r in rows if r[2] > 2)} / {len(rows)}")
        for r in rows[:10]:
            print("     hip %.2e ref %.2e ratio %6.1f max|g| %.2e %s" % r)
