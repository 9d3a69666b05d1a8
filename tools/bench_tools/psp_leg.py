"""one bench_psp.train_leg (argv: decoder dtype, size, batch, train_decoder 0/1, [graph], [script]) for rocprofv3 --kernel-trace
--stats: wall time per step printed, so that the kernel-time sum of the trace shows how much of the step is launch-bound"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench_psp as B
dd, size, batch, td = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
graph = "graph" in sys.argv[5:]
la = B.SCRIPT_LOSS_ARGS if "script" in sys.argv[5:] else None
steps = 6
ed = "bf16" if "enc16" in sys.argv[5:] else "fp32"
dt, summ = B.train_leg(torch.device("cuda:0"), dd, size, batch, steps, 2, train_decoder=td, graph=graph, loss_args=la, encoder_dtype=ed)
print(f"{dd} {size} bs{batch} train_decoder={td} graph={graph} script_loss={la is not None} encoder={ed}: {dt / steps * 1e3:.1f} ms per step (wall), {batch * steps / dt:.1f} images/s; profiled step: {summ}")
if "table" in sys.argv[5:]:  # per-call-site launch table of one eager step
    from collections import defaultdict
    from face_mask_inpaint_amd import functional as FF
    import types
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from face_mask_inpaint_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    opts = types.SimpleNamespace(output_size=size, encoder_type="GradualStyleEncoder", train_decoder=td, use_attention=True, pt_ckpt_path=None,
                                 stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype=dd)
    net = pSp(opts).to(dev).train()
    net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
    crit = pSpLoss(types.SimpleNamespace(**(la or B.LOSS_ARGS))).to(dev)
    opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    x, ref, y, m = B.synth(batch, dev)
    def step():
        y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
        loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
        opt.zero_grad(); loss.backward(); opt.step()
    step(); step(); torch.cuda.synchronize()
    FF.PROFILE = []
    step(); torch.cuda.synchronize()
    recs, FF.PROFILE = FF.PROFILE, None
    agg = defaultdict(lambda: [0.0, 0, 0.0])
    for t, f, s, e in recs:
        a = agg[t]; a[0] += s.elapsed_time(e); a[1] += 1; a[2] += f
    tot = sum(a[0] for a in agg.values())
    print("profiled launches: %.1f ms" % tot)
    for t, (ms, n, f) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
        print(f"{ms:8.3f} ms {n:4d} launches {f / ms / 1e9 if not t.startswith('bytes:') else 0:7.1f} TF  {t}")
