#!/usr/bin/env python
"""Data-parallel train_psp step (BASELINE.json configs[4]: pSp with the 1024^2 StyleGAN2 decoder in bf16, 4 images per GPU) -- NOT the
driver's bench (bench.py).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_psp_dp.py --gpus N

Each rank holds a full replica; the encoder's 210 M fp32 gradients (840 MB) are all-reduced in 16 MB buckets launched from
post-accumulate hooks while the backward pass is still producing the earlier layers' gradients (face_mask_inpaint_amd/distributed.py).
BatchNorm statistics stay per GPU, as a torch DDP run of the reference would keep them.  Prints ONE JSON line (rank 0):
whole-job images/s, MAX over ranks of the timed region.  `FMI_REHEARSAL_ONE_GPU=1` runs all ranks on cuda:0 over gloo (rehearsal of the
multi-process path on a one-GPU box)."""
import argparse
import json
import os
import time
import types

import torch

import bench_psp as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--size", type=int, default=1024, help="decoder output size (256 or 1024)")
    ap.add_argument("--decoder-dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from face_mask_inpaint_amd import launch

    if launch.needs_spawn(args.gpus):  # started without a launcher: one child process per GPU, before this process touches the GPU
        import sys

        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = bool(os.environ.get("FMI_REHEARSAL_ONE_GPU"))
    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo" if rehearsal else "nccl")
    from face_mask_inpaint_amd.distributed import DataParallelOptimizer, broadcast_parameters
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from face_mask_inpaint_amd.optim import FusedAdam

    torch.manual_seed(0)
    opts = types.SimpleNamespace(output_size=args.size, encoder_type="GradualStyleEncoder", train_decoder=False, use_attention=True,
                                 pt_ckpt_path=None, stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True,
                                 decoder_dtype=args.decoder_dtype)
    net = pSp(opts).to(dev).train()
    net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
    broadcast_parameters([net])
    crit = pSpLoss(types.SimpleNamespace(id_lambda=0, lpips_lambda=0, l2_lambda=1.0, style_lambda=0, lpips_lambda_ref=0, l2_lambda_ref=1.0,
                                         cx_lambda=0, w_norm_lambda=0.005, start_from_latent_avg=True))
    opt = FusedAdam([p for p in net.encoder.parameters() if p.requires_grad], lr=1e-4)
    if world > 1:
        opt = DataParallelOptimizer(opt)
    x, ref, y, m = B.synth(args.batch, dev, seed=1234 + rank)

    def step():
        y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
        loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def sync():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).item()
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # replicas must still agree after the averaged updates
        p = next(net.encoder.parameters()).detach().float().cpu().flatten()[:64].clone()
        q = p.clone()
        dist.broadcast(q, 0)
        assert torch.equal(p, q), "replicas diverged"
    if rank == 0:
        print(json.dumps({"metric": "train_psp images/sec (data parallel)", "value": round(world * args.batch * args.steps / dt, 2), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "scaling": "weak", "higher_is_better": True, "data": "synthetic",
                          "config": {"workload": f"train_psp RefpSp + attention, StyleGAN2 {args.size}^2 decoder ({args.decoder_dtype}), encoder fp32",
                                     "batch_per_gpu": args.batch, "parallelism": f"dp{world}",
                                     "collectives_per_step": getattr(opt, "collectives", 0) // max(args.steps + args.warmup, 1)}}), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
