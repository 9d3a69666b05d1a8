"""where does the end-to-end gradient error enter?  d(loss term)/d(gen) and d<gen, r>/d(params) of the tiny golden config:
HIP vs the CPU oracle evaluated in float64, next to the oracle's own fp32 evaluation (argv[1] = step, default 0)"""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF
from oracle import picnet_cpu as O
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 0
f2 = dict(fx); f2["G_sd0"], f2["D_sd0"] = fx[f"G_sd{step}"], fx[f"D_sd{step}"]
s = fx[f"step{step}"]; cfg = fx["config"]
G, D, gopt, optG, optD = T._tiny_models(f2, dev)
m = FF.binarise_mask(s["mask"].to(dev))
kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"], dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)

def oracle(dt):
    c = lambda t: t.to(dt) if t.is_floating_point() else t
    PG = O.prepare_params(f2["G_sd0"], dtype=dt)
    PD = O.prepare_params(f2["D_sd0"], frozen=True, dtype=dt)
    PV = O.prepare_params(fx["V_sd"], frozen=True, dtype=dt)
    return PG, PD, PV, c

gen32 = s["gen"]  # the reference's fp32 image: the common evaluation point of every loss-term gradient below
res = {}
for name, dt in (("o64", torch.float64), ("o32", torch.float32)):
    PG, PD, PV, c = oracle(dt)
    mask = c(O.binarise_mask(s["mask"]))
    src, gt, ref = c(s["src"]), c(s["gt"]), c(s["ref"])
    terms = {
        "perceptual": lambda g: O.vgg_loss(PV, "", g, gt, "perceptual"),
        "style": lambda g: O.vgg_loss(PV, "", g * (1 - mask).unsqueeze(1), src, "style"),
        "contextual": lambda g: O.vgg_loss(PV, "", g * mask.unsqueeze(1), ref * mask.unsqueeze(1), "contextual"),
        "l1": lambda g: F.l1_loss(g, gt),
        "gan": lambda g: O.lsgan(O.res_discriminator(PD, "", g, cfg["disc_layers"]), True),
    }
    out = {}
    for k, fn in terms.items():
        PD = O.prepare_params(f2["D_sd0"], frozen=True, dtype=dt)
        g = c(gen32).clone().requires_grad_(True)
        l = fn(g); l.backward()
        out[k] = (float(l), g.grad.double())
    r = torch.randn(gen32.shape, generator=torch.Generator().manual_seed(1))
    gen = O.reference_fill_forward(PG, src, ref, mask, c(s["eps_p"]), c(s["eps_q"]), **kw)
    (gen * c(r)).sum().backward()
    out["_gen"] = gen.detach().double()
    out["_params"] = {n: PG[n].grad.double() for n, _ in G.named_parameters() if PG[n].grad is not None}
    res[name] = out

src, gt, ref = s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev)
import copy
d_sd = copy.deepcopy(D.state_dict())
hterms = {
    "perceptual": lambda g: gopt.vgg_loss(g, gt, lossType="perceptual"),
    "style": lambda g: gopt.style_loss(g, src, m),
    "contextual": lambda g: gopt.contextual_loss(g, ref, m),
    "l1": lambda g: FF.l1_loss(FF.to_nhwc(g), FF.to_nhwc(gt)),
    "gan": lambda g: gopt.gan_loss(D(g), True, False),
}
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-300)), float((a - b).norm() / (b.norm() + 1e-300))
for k, fn in hterms.items():
    D.load_state_dict(d_sd)
    gopt.vgg_loss._cache = {}
    g = gen32.to(dev).clone().requires_grad_(True)
    l = fn(g); l.backward()
    gh = g.grad.detach().cpu().double()
    l64, g64 = res["o64"][k]; l32, g32 = res["o32"][k]
    print("%-11s loss rel err hip %.2e o32 %.2e | grad max-rel hip %.2e o32 %.2e | l2-rel hip %.2e o32 %.2e" % (
        k, abs(float(l) / l64 - 1), abs(l32 / l64 - 1), rel(gh, g64)[0], rel(g32, g64)[0], rel(gh, g64)[1], rel(g32, g64)[1]))
G.zero_grad()
gen = G(src, ref, src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
(gen * r.to(dev)).sum().backward()
print("gen fwd max err: hip %.2e o32 %.2e" % (float((gen.detach().cpu().double() - res["o64"]["_gen"]).abs().max()), float((res["o32"]["_gen"] - res["o64"]["_gen"]).abs().max())))
rows = []
for n, g64 in res["o64"]["_params"].items():
    p = dict(G.named_parameters())[n]
    if p.grad is None or float(g64.abs().max()) < 1e-12: continue
    rows.append((rel(p.grad.detach().cpu().double(), g64)[0], rel(res["o32"]["_params"][n], g64)[0], n))
rows.sort(reverse=True)
print("generator backward with a fixed cotangent: worst hip %.2e, worst o32 %.2e" % (rows[0][0], max(r[1] for r in rows)))
for r_ in rows[:8]: print("   hip %.2e o32 %.2e %s" % r_)
