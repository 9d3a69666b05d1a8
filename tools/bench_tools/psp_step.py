"""train_psp step (bench_psp.py's leg) for rocprofv3 --kernel-trace --stats; argv[1] = fp32 | bf16 decoder"""
import sys, time, types
import torch
sys.path.insert(0, "/root/repo")
import bench_psp as B
from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
from face_mask_inpaint_amd.modules.psp.psp import pSp
from face_mask_inpaint_amd.optim import FusedAdam
dev = torch.device("cuda:0")
dd = sys.argv[1] if len(sys.argv) > 1 else "fp32"
torch.manual_seed(0)
opts = types.SimpleNamespace(output_size=256, encoder_type="GradualStyleEncoder", train_decoder=False, use_attention=True, pt_ckpt_path=None,
                             stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype=dd)
net = pSp(opts).to(dev).train()
net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
crit = pSpLoss(types.SimpleNamespace(id_lambda=0, lpips_lambda=0, l2_lambda=1.0, style_lambda=0, lpips_lambda_ref=0, l2_lambda_ref=1.0, cx_lambda=0,
                                     w_norm_lambda=0.005, start_from_latent_avg=True))
opt = FusedAdam([p for p in net.encoder.parameters() if p.requires_grad], lr=1e-4)
x, ref, y, m = B.synth(16, dev)
def step():
    y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
    loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    step()
torch.cuda.synchronize()
print(f"{dd}: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per step")
if len(sys.argv) > 2:  # per-call-site table
    from face_mask_inpaint_amd import functional as FF
    from collections import defaultdict
    FF.PROFILE = []
    step()
    torch.cuda.synchronize()
    recs, FF.PROFILE = FF.PROFILE, None
    agg = defaultdict(lambda: [0.0, 0, 0.0])
    for t, f, s, e in recs:
        a = agg[t]
        a[0] += s.elapsed_time(e); a[1] += 1; a[2] += f
    for t, (ms, n, f) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{ms:8.3f} ms {n:4d} launches {f / ms / 1e9 if not t.startswith('bytes:') else 0:7.1f} TF  {t}")
