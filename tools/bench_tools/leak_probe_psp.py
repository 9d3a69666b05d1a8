"""census of live CUDA tensors across pSp training steps (fp32 and bf16 decoder)"""
import gc, sys, types, collections
import torch
sys.path.insert(0, "/root/repo")
import bench_psp as B
from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
from face_mask_inpaint_amd.modules.psp.psp import pSp
from face_mask_inpaint_amd.optim import FusedAdam
dev = torch.device("cuda:0")
for dd in ("bf16",):
    opts = types.SimpleNamespace(output_size=256, encoder_type="GradualStyleEncoder", train_decoder=True, use_attention=True, pt_ckpt_path=None,
                                 stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype=dd)
    net = pSp(opts).to(dev).train()
    net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
    crit = pSpLoss(types.SimpleNamespace(id_lambda=0, lpips_lambda=0, l2_lambda=1.0, style_lambda=0, lpips_lambda_ref=0, l2_lambda_ref=1.0, cx_lambda=0,
                                         w_norm_lambda=0.005, start_from_latent_avg=True))
    opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    x, ref, y, m = B.synth(4, dev)
    def census():
        c = collections.Counter()
        for o in gc.get_objects():
            try:
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    c[(tuple(o.shape), str(o.dtype))] += 1
            except Exception:
                pass
        return c
    out = []
    for i in range(14):
        y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
        loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
        opt.zero_grad(); loss.backward(); opt.step()
        if i in (5, 13):
            torch.cuda.synchronize(); gc.collect(); c = census(); out.append(c)
            print(dd, i, "alloc GB", round(torch.cuda.memory_allocated() / 2**30, 3), "tensors", sum(c.values()))
    diff = {k: out[1][k] - out[0].get(k, 0) for k in out[1] if out[1][k] != out[0].get(k, 0)}
    for k, v in sorted(diff.items(), key=lambda kv: -abs(kv[1]))[:10]:
        print(v, k)
