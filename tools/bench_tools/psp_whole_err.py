"""per-tensor gradient error of the whole-pSp training-mode fixture (tests/golden/psp_whole.pt): HIP vs the reference's float64
digests, next to the reference's own fp32 error; two runs to show run-to-run variation"""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from face_mask_inpaint_amd.modules.psp.psp import pSp
from oracle.seeded import digest_error, seeded_fill_, seeded_tensor
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/psp_whole.pt"), weights_only=True)
cfg = fx["config"]
opts = types.SimpleNamespace(output_size=cfg["output_size"], encoder_type="GradualStyleEncoder", use_attention=True, train_decoder=True,
                             start_from_latent_avg=True, learn_in_w=False, pt_ckpt_path=None, stylegan_weights=None)
net = pSp(opts)
seeded_fill_(net, cfg["seed"])
net.latent_avg = seeded_tensor((opts.n_styles, 512), cfg["latent_avg_seed"], 0.5)
net.to(dev).train()
mask = torch.zeros(2, 256, 256)
for i, (a, b, c, d) in enumerate(cfg["rects"]):
    mask[i, a:b, c:d] = 1
mask = mask.to(dev)
prev = None
for rep in range(2):
    net.zero_grad()
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["x_seed"])) * 2 - 1).to(dev).requires_grad_(True)
    ref = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["ref_seed"])) * 2 - 1).to(dev).requires_grad_(True)
    img, lat = net(x, ref=ref, src_mask=mask, resize=True, randomize_noise=False, return_latents=True)
    ((img * seeded_tensor(img.shape, cfg["cot_seeds"][0]).to(dev)).sum() / 256.0 + (lat * seeded_tensor(lat.shape, cfg["cot_seeds"][1]).to(dev)).sum()).backward()
    P = dict(net.named_parameters())
    rows = []
    for n, d in fx["gparams64"].items():
        if float(d["max"]) <= 1e-20: continue
        e = digest_error(P[n].grad, d)
        r = float((fx["gparams"][n]["sample"] - d["sample"]).abs().max()) / float(d["max"])
        rows.append((e / max(r, 1e-7), e, r, P[n].grad.ndim, n))
    rows.sort(reverse=True)
    print(f"run {rep}: img err {float((img.detach().cpu() - fx['image']).abs().max()):.2e}  gx {digest_error(x.grad, fx['gx64']):.2e} gref {digest_error(ref.grad, fx['gref64']):.2e}")
    for r_ in rows[:25]:
        print("   ratio %8.1f hip %.2e ref %.2e ndim %d %s" % r_)
    cur = {n: P[n].grad.clone() for n in P if P[n].grad is not None}
    if prev is not None:
        dd = sorted(((float((cur[n] - prev[n]).abs().max() / (prev[n].abs().max() + 1e-30)), n) for n in cur), reverse=True)
        print("run-to-run relative differences:", [("%.1e" % a, b) for a, b in dd[:5]])
    prev = cur
