"""fill / add / copy launches of one train_psp step grouped by tensor shape (torch profiler)"""
import sys, types, collections
import torch
sys.path.insert(0, "/root/repo")
import bench_psp as B
from torch.profiler import profile, ProfilerActivity
from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
from face_mask_inpaint_amd.modules.psp.psp import pSp
from face_mask_inpaint_amd.optim import FusedAdam
dev = torch.device("cuda:0")
opts = types.SimpleNamespace(output_size=256, encoder_type="GradualStyleEncoder", train_decoder=False, use_attention=True, pt_ckpt_path=None,
                             stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype="bf16")
net = pSp(opts).to(dev).train()
net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
crit = pSpLoss(types.SimpleNamespace(**(B.SCRIPT_LOSS_ARGS if "script" in sys.argv else B.LOSS_ARGS)))
if hasattr(crit, "to"):
    crit = crit.to(dev)
opt = FusedAdam([p for p in net.encoder.parameters() if p.requires_grad], lr=1e-4)
x, ref, y, m = B.synth(16, dev)
def step():
    y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
    loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat", "aten::sum", "aten::to"):
        cnt[(ev.name, str(ev.input_shapes)[:90])] += 1
for (name, shp), c in cnt.most_common(36):
    print(f"{c:5d}  {name:12s} {shp}")
