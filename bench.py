#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the PICNet-ref training step (train_reference_fill.py:331-346 +
GANOptimizer.__call__, loss.py:120-134) at 256x256, bs = 8 per GPU, fp32, synthetic CelebA-HQ-shaped data
(BASELINE.json configs[1]; configs[3] when launched on 8 GPUs).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline:     achieved TFLOP/s (algorithmic fp32 FLOPs) of the dominant kernel and of the whole matrix-core family (csrc/gemm_core.h,
                conv3x3.h, attention.hip), measured with events on the launch stream around every launch of one extra (untimed)
                training step.  The family computes every fp32 product as SIX bf16 MFMAs on exact three-way bf16 splits of both
                operands (csrc/x6.h), so its ceiling is the dense bf16 matrix peak / 6 = 419.4 TFLOP/s of fp32-equivalent work
                (the 157.3 TFLOP/s fp32-MFMA peak the round-1 kernels were priced against is reported next to it);
  cpu_baseline: the CPU restatement (oracle/picnet_cpu.py, kind "port") timed on this host's cores on a bounded
                sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate
BF16_MFMA_PEAK_TFLOPS = 2516.6  # "Peak BF16/FP16 MFMA" ~2.5 PF dense = 16 x the fp32 matrix rate
X6_PRODUCTS = 6                # bf16 MFMAs per fp32 product (csrc/x6.h)
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / X6_PRODUCTS  # 419.4: the most fp32-equivalent work the scheme can deliver
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def pmc_traffic(call_site_key):
    """HBM-side bytes per launch of the kernel behind ``call_site_key`` from the rocprofv3 PMC passes committed under profiles/
    (pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KB per launch; FETCH_SIZE doubled as the guide
    prescribes for 16-byte-per-lane streaming reads on gfx950; WRITE_SIZE counts fp32 atomics exactly).  The entry names the kernel
    it was collected on and the sha256 of the kernel's source file at that time: when the source has changed since, the number is
    STALE and is not reported (traffic = null plus the reason; tests/test_host_modules.py fails on the same condition)."""
    import hashlib

    try:
        tab = json.load(open(PMC_TRAFFIC_FILE))
    except Exception as e:  # noqa: BLE001
        return None, {"traffic_error": f"profiles/pmc_traffic.json unreadable: {e}"}
    ent = tab.get(call_site_key)
    if ent is None:
        return None, {"traffic_error": f"no PMC profile for {call_site_key!r} in profiles/pmc_traffic.json"}
    src = os.path.join(ROOT, ent["source"])
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    if sha != ent["source_sha256"]:
        print(f"[bench] PMC traffic for {call_site_key!r} is STALE: {ent['source']} changed since {ent['profile']} was collected",
              file=sys.stderr, flush=True)
        return None, {"traffic_error": f"stale: {ent['source']} changed since {ent['profile']} was collected on {ent['kernel']}"}
    return 2.0 * ent["fetch_kb"] * 1024 + ent["write_kb"] * 1024, {
        "traffic_unit": "bytes per launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, %s)" % ent["profile"], "traffic_kernel": ent["kernel"]}


ENC = dict(type="pluralistic", ngf=32, z_nc=128, img_f=128, layers=5, norm="none", activation="LeakyReLU", L=6)
DEC = dict(ngf=32, z_nc=256, img_f=256, layers=5, norm="instance", activation="LeakyReLU", L=0)
DISC = dict(ndf=32, img_f=128, layers=5, norm="none", activation="LeakyReLU", model_type="ResDis")
LR = 1e-5  # train_reference_fill.py:22 default


def build_models(dev, world):
    from face_mask_inpaint_amd import distributed as fdist
    from face_mask_inpaint_amd.modules.loss import GANOptimizer
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from face_mask_inpaint_amd.modules.pluralistic_model import network
    from face_mask_inpaint_amd.optim import FusedAdam

    torch.manual_seed(0)  # identical initial weights on every rank
    G = ReferenceFill(None, dict(ENC), dict(DEC), use_att=True, out_size=(256, 256)).to(dev)
    D = network.define_d(**DISC).to(dev)
    optG = FusedAdam([p for p in G.parameters() if p.requires_grad], lr=LR)
    optD = FusedAdam([p for p in D.parameters() if p.requires_grad], lr=LR)
    if world > 1:
        fdist.broadcast_parameters([G, D])
        optG, optD = fdist.DataParallelOptimizer(optG), fdist.DataParallelOptimizer(optD)
    gopt = GANOptimizer(optD, optG).to(dev)
    if world > 1:
        fdist.broadcast_parameters([gopt.vgg_loss])
    return G, D, gopt


def synthetic(n, size, seed, dev):
    """U[0,1) images, int64 binary_map (0 / 255 inside a random lower-face ellipse)"""
    g = torch.Generator().manual_seed(seed)
    src, ref, gt = (torch.rand(n, 3, size, size, generator=g) for _ in range(3))
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    s = size / 256.0
    mask = torch.zeros(n, size, size, dtype=torch.long)
    for i in range(n):
        r = torch.rand(4, generator=g)
        cy, cx = (176 + r[0] * 32 - 16) * s, (128 + r[1] * 32 - 16) * s
        ry, rx = (56 + r[2] * 24 - 12) * s, (80 + r[3] * 24 - 12) * s
        mask[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
    return [t.to(dev) for t in (src, ref, gt, mask)]


def train_step(G, D, gopt, batch):
    from face_mask_inpaint_amd import functional as FF

    src, ref, gt, mask = batch
    m = FF.binarise_mask(mask)                      # train_reference_fill.py:340
    gen = G(src, ref, src_mask=m)                   # :342 (fresh N(0,1) draws for rsample)
    return gopt(D, src, gt, ref, gen, m)            # :344-346


def host_cores() -> int:
    """cores this process may actually use: min(affinity, cgroup cpu quota) -- the GPU box shows 256 logical CPUs
    but grants a 16-CPU quota, and 256 torch threads on 16 CPUs never finish"""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(size, seconds_budget=40.0):
    """the CPU port of the same step on this host's cores (SURVEY.md 8d protocol): fp32, all cores the cgroup grants, bs = 1 (no
    batch-coupled op on the path), 2 warm-up + >= 5 timed steps, median; plus ONE single-thread step.  Bounded: the timed loop stops
    after ``seconds_budget`` seconds once 3 steps are in, the single-thread step is skipped when a step would take > 90 s."""
    from oracle import picnet_cpu as O  # checker / baseline only -- never imported by the product package
    from face_mask_inpaint_amd.modules.loss import VGGLoss
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from face_mask_inpaint_amd.modules.pluralistic_model import network

    torch.manual_seed(0)
    cores = host_cores()
    torch.set_num_threads(cores)
    G = ReferenceFill(None, dict(ENC), dict(DEC), use_att=True, out_size=(256, 256))
    D = network.define_d(**DISC)
    PG, PD = O.prepare_params(G.state_dict()), O.prepare_params(D.state_dict())
    PV = O.prepare_params(VGGLoss().state_dict(), frozen=True)
    og = torch.optim.Adam(O.unique_trainable(PG), lr=LR)
    od = torch.optim.Adam(O.unique_trainable(PD), lr=LR)
    src, ref, gt, mask, eps_p, eps_q = O.synthetic_batch(1, size, seed=1234, feat_hw=size // 8, z_nc=128)
    # parity of the generated image on the same weights / inputs: HIP path vs this CPU path (before any update)
    parity = None
    if torch.cuda.is_available():
        import copy

        from face_mask_inpaint_amd import functional as FF
        from face_mask_inpaint_amd.modules.evaluations.ssim import ssim as ssim_hip

        dev = torch.device("cuda", torch.cuda.current_device())
        Gd = copy.deepcopy(G).to(dev)
        with torch.no_grad():
            got = Gd(src.to(dev), ref.to(dev), src_mask=FF.binarise_mask(mask.to(dev)), eps=(eps_p.to(dev), eps_q.to(dev)))
            want = O.reference_fill_forward(O.prepare_params(G.state_dict()), src, ref, O.binarise_mask(mask), eps_p, eps_q, out_size=(size, size))
            wd = want.to(dev).contiguous()
            parity = {"ssim_vs_cpu": round(float(ssim_hip(got.contiguous(), wd)), 6),
                      "max_abs_err": float((got - wd).abs().max()), "max_rel_to_range": float((got - wd).abs().max() / wd.abs().max()),
                      "mask_bit_exact": bool(torch.equal(FF.binarise_mask(mask.to(dev)).cpu(), O.binarise_mask(mask)))}
        del Gd
    WARM = 2
    times = []
    t_all = time.time()
    for it in range(WARM + 5):
        t0 = time.time()
        O.train_step(PG, PD, PV, og, od, src, gt, ref, mask, eps_p, eps_q, out_size=(size, size))
        times.append(time.time() - t0)
        if time.time() - t_all > seconds_budget and it >= WARM + 2:
            break
    timed = times[WARM:]
    sec = sorted(timed)[len(timed) // 2]
    one = None
    if sec * cores * 0.7 < 90.0:  # a single-thread step is at most ~cores x slower
        torch.set_num_threads(1)
        t0 = time.time()
        O.train_step(PG, PD, PV, og, od, src, gt, ref, mask, eps_p, eps_q, out_size=(size, size))
        one = round(1.0 / (time.time() - t0), 4)
        torch.set_num_threads(cores)
    return {"value": round(1.0 / sec, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "single_thread_value": one, "parity": parity,
            "sample": "oracle/picnet_cpu.py train_step at %dx%d, fp32, bs=1 (path has no batch-coupled op), %d warm-up + %d timed steps, median; "
                      "single_thread_value = one step with torch.set_num_threads(1)" % (size, size, WARM, len(timed))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C5 pSp legs appended as 'extra' (N = 1 only)")
    ap.add_argument("--profile-dump", default=None, help="write the per-shape launch table of the profiled step here")
    args = ap.parse_args()

    from face_mask_inpaint_amd import launch

    if launch.needs_spawn(args.gpus):  # started without a launcher: one child process per GPU, before this process touches the GPU
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU fallback)"
    if os.environ.get("FMI_REHEARSAL_ONE_GPU"):  # 2-rank rehearsal of the multi-process path on a one-GPU box (gloo, all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        if os.environ.get("FMI_REHEARSAL_ONE_GPU"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" = RCCL over xGMI; device_id binds the communicator to this rank's GPU
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from face_mask_inpaint_amd import functional as FF

    G, D, gopt = build_models(dev, world)
    batch = synthetic(args.batch, args.size, 1234 + rank, dev)

    def sync():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(G, D, gopt, batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = train_step(G, D, gopt, batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(l).item() for l in losses), "non-finite loss"

    roofline = None
    if not args.no_roofline and rank != 0:
        train_step(G, D, gopt, batch)  # the profiled step is a collective step: every rank takes part, rank 0 records
        torch.cuda.synchronize()
    if not args.no_roofline and rank == 0:
        FF.PROFILE = []
        train_step(G, D, gopt, batch)
        torch.cuda.synchronize()
        recs, FF.PROFILE = FF.PROFILE, None
        recs = [r for r in recs if not r[0].startswith("bytes:")]  # bandwidth-kernel records (pSp path) carry bytes, not flops
        tot_ms = sum(s.elapsed_time(e) for _, _, s, e in recs)
        tot_fl = sum(f for _, f, _, _ in recs)
        by, detail = {}, {}
        for tag, f, s, e in recs:
            d = detail.setdefault(tag, [0.0, 0.0, 0])
            d[0] += f
            d[1] += s.elapsed_time(e)
            d[2] += 1
            tag = tag.split("|")[0]
            a = by.setdefault(tag, [0.0, 0.0, 0])
            a[0] += f
            a[1] += s.elapsed_time(e)
            a[2] += 1
        if args.profile_dump:
            with open(args.profile_dump, "w") as fh:
                for k, v in sorted(detail.items(), key=lambda kv: -kv[1][1]):
                    fh.write("%9.3f ms %4d launches %7.1f TFLOP/s  %s\n" % (v[1], v[2], v[0] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0, k))
        top = sorted(by.items(), key=lambda kv: -kv[1][1])
        fam = tot_fl / (tot_ms * 1e-3) / 1e12
        dom_key, (dom_fl, dom_ms, dom_n) = sorted(detail.items(), key=lambda kv: -kv[1][1])[0]  # one kernel at one shape
        dom_tag = dom_key.split("|")[0]
        ach = dom_fl / (dom_ms * 1e-3) / 1e12
        kernel_names = {"attn_fused_bwd": "attn_bwd2_x6_kernel<64,8> (csrc/attention.hip: fused softmax(QQ^T)V backward, fp32 products as 6 bf16 MFMAs)",
                        "attn_fused_fwd": "attn_fwd_x6_kernel<64,8,8> (csrc/attention.hip)"}
        traffic, traffic_note = pmc_traffic(dom_key)
        # SURVEY.md 8(d) prices the attention backward at 2 x the forward's 2 N T^2 (d + C); the flash-style backward also RECOMPUTES
        # the q q^T product (2 N T^2 (3 d + 2 C) executed): "frac" uses the 8(d) count, "frac_with_recompute" the executed one
        survey_fl = dom_fl
        if dom_tag == "attn_fused_bwd":
            import re

            mm = re.match(r"T(\d+) d(\d+) C(\d+) b(\d+)", dom_key.split("|")[-1])
            t_, d_, c_, b_ = (int(v) for v in mm.groups())
            survey_fl = dom_n * 4.0 * b_ * t_ * t_ * (d_ + c_)
        ach8d = survey_fl / (dom_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach8d, 2), "peak": round(X6_PEAK_TFLOPS, 1), "unit": "TFLOP/s",
                    "frac": round(ach8d / X6_PEAK_TFLOPS, 4), "traffic": traffic, **traffic_note,
                    "peak_note": "fp32 products as 6 bf16 MFMAs on exact 3-way bf16 splits (csrc/x6.h): peak = %.1f TFLOP/s dense bf16 / 6; achieved counts ALGORITHMIC fp32 FLOPs, "
                                 "the matrix pipe executes 6 x as many bf16 FLOPs" % BF16_MFMA_PEAK_TFLOPS,
                    "executed_bf16_tflops": round(X6_PRODUCTS * ach, 1), "frac_of_bf16_peak_executed": round(X6_PRODUCTS * ach / BF16_MFMA_PEAK_TFLOPS, 4),
                    "vs_fp32_mfma_peak": round(ach8d / FP32_MFMA_PEAK_TFLOPS, 4),
                    "flop_count": "SURVEY.md 8(d): backward = 2 x forward = 4 N T^2 (d + C)" if survey_fl != dom_fl else "2 M N K per launch",
                    "achieved_with_recompute": round(ach, 2), "frac_with_recompute": round(ach / X6_PEAK_TFLOPS, 4),
                    "kernel": kernel_names.get(dom_tag, "gemm_mfma_f32_kernel<...> call site " + dom_tag), "shape": dom_key.split("|")[-1],
                    "launches_per_step": dom_n, "avg_launch_ms": round(dom_ms / dom_n, 3),
                    "algorithmic_tflop_per_launch": round(survey_fl / dom_n / 1e12, 4),
                    "mfma_family": {"achieved": round(fam, 2), "frac": round(fam / X6_PEAK_TFLOPS, 4), "vs_fp32_mfma_peak": round(fam / FP32_MFMA_PEAK_TFLOPS, 4), "launches": len(recs),
                                    "kernel_ms_per_step": round(tot_ms, 2), "algorithmic_tflop_per_step": round(tot_fl / 1e12, 3)},
                    "by_call_site": {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else None, "ms": round(v[1], 2), "launches": v[2]}
                                     for k, v in top[:7]}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.size)
    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        # BASELINE configs[2] (C3) and the single-GPU leg of configs[4] (C5), so that the driver's fixed command times them too
        del G, D, gopt, batch
        torch.cuda.empty_cache()
        import bench_psp

        extra = bench_psp.extra_block(dev)
    collectives = None
    if world > 1:
        import torch.distributed as dist

        mine = torch.tensor([getattr(o, "collectives", 0) for o in (gopt.optimizer_G, gopt.optimizer_D)], device=dev, dtype=torch.int64)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        collectives = {"per_rank_G_D_over_%d_steps" % (args.steps + args.warmup + (0 if args.no_roofline else 1)): [c.tolist() for c in allc]}

    if rank == 0:
        imgs = args.batch * world * args.steps
        out = {"metric": "train images/sec at 256x256 bs=8 per GPU (PICNet-ref train step)", "value": round(imgs / dt, 3),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "train_reference_fill.py PICNet-ref %dx%d fp32, bs=%d per GPU, synthetic CelebA-HQ-shaped batch + random binary_map (BASELINE configs[1]%s)"
                                      % (args.size, args.size, args.batch, "; configs[3] = bs 64 over 8 GPUs" if world == 8 else ""),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world},
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        if collectives is not None:
            import torch.distributed as dist
            from face_mask_inpaint_amd import distributed as fdist

            out["collectives"] = collectives
            out["process_group"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "exchange": fdist.default_exchange(),
                                    "launcher": "self-spawned" if os.environ.get("FMI_SELF_SPAWNED") else "external"}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
