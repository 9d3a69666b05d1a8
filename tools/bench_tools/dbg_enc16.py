import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench_psp as B
from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
from face_mask_inpaint_amd.modules.psp.psp import pSp
from face_mask_inpaint_amd.optim import FusedAdam
dev = torch.device("cuda:0")
td = bool(int(sys.argv[1])); ed = sys.argv[2]
torch.manual_seed(0)
opts = types.SimpleNamespace(output_size=256, encoder_type="GradualStyleEncoder", train_decoder=td, use_attention=True, pt_ckpt_path=None,
                             stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype="bf16", encoder_dtype=ed)
net = pSp(opts).to(dev).train()
net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
crit = pSpLoss(types.SimpleNamespace(**B.LOSS_ARGS)).to(dev)
params = [p for p in net.encoder.parameters() if p.requires_grad] + ([p for p in net.decoder.parameters() if p.requires_grad] if td else [])
opt = FusedAdam(params, lr=1e-4)
x, ref, y, m = B.synth(16, dev)
for it in range(10):
    y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
    loss, ld, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
    opt.zero_grad(); loss.backward()
    gn = max(float(p.grad.abs().max()) for p in params if p.grad is not None)
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    opt.step()
    print(it, {k: round(v, 5) for k, v in ld.items()}, "latent absmax %.3g" % float(latent.abs().max()), "yhat absmax %.3g" % float(y_hat.float().abs().max()), "max|grad| %.3g" % gn, bad[:3])
