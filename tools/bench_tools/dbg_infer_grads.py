"""per-parameter deviation of the ReferenceFill 'pool' variant gradients from the fixture digests (tests/test_gpu_infer.py), for the
library FMI_LIB_PATH selects"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd.modules.model import ReferenceFill
from oracle.seeded import digest_error

ENC = dict(type="pluralistic", ngf=8, z_nc=8, img_f=16, layers=5, norm="none", activation="LeakyReLU", L=2)
DEC = dict(ngf=8, z_nc=16, img_f=32, layers=5, norm="instance", activation="LeakyReLU", L=0)
dev = torch.device("cuda:0")
f = torch.load(os.path.join(os.path.dirname(__file__), "../../tests/golden/picnet_infer.pt"), weights_only=True)["variants"]
G = ReferenceFill(None, dict(ENC), dict(DEC), use_att=True, out_size=(100, 90))
G.load_state_dict(f["sd0"], strict=False)
G = G.to(dev)
src, ref, mask = f["src"].to(dev), f["ref"].to(dev), f["mask"].to(dev)
with torch.no_grad():  # the fixture's sequence: the spectral-norm u / v advance with every forward
    G(src, ref, src_mask=mask, no_prior=True)
    G(src, ref, src_mask=mask, resize=False, eps=tuple(e.to(dev) for e in f["raw_eps"]))
o = G(src, ref, src_mask=mask, eps=tuple(e.to(dev) for e in f["pool_eps"]))
print("out err", float((o.detach().cpu() - f["pool"]).abs().max()), "of", float(f["pool"].abs().max()))
(o * f["gout"].to(dev)).sum().backward()
P = dict(G.named_parameters())
errs = sorted(((digest_error(P[n].grad, d), n) for n, d in f["gparams"].items() if float(d["max"]) > 1e-5), reverse=True)
for e, n in errs[:12]:
    print("%.3e %s" % (e, n))
print("median %.3e" % errs[len(errs) // 2][0])
