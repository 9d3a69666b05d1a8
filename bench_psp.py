#!/usr/bin/env python
"""Secondary measurements for the pSp path (BASELINE.json configs[2] / configs[4]; SURVEY.md 8d targets) -- NOT the driver's
bench (that is bench.py, the PICNet-ref training step).  One MI355X, fp32, synthetic data, random-init weights:

  * train_psp step (GradualStyleEncoder IR-SE50 on src + ref, example-guided attention, StyleGAN2 256^2 decoder, masked-L2 +
    reference-L2 + W-norm loss, fused Adam), bs 16: images/s;
  * ModulatedConv2d: MFMA utilisation of every convolution launch inside Generator forward / backward (north_star target:
    >= 50 % MFMA utilisation), for the fp32 decoder against the 157.3 TFLOP/s fp32 matrix peak AND for the bf16 decoder of
    configs[2] ("bf16_decoder": bf16 activations, fp32 accumulate) against the 2516.6 TFLOP/s dense bf16 peak;
  * upfirdn2d / fused noise+bias+lrelu: achieved GB/s (algorithmic bytes = in + out) against the 8 TB/s HBM roofline, for the
    256^2 decoder and for the 1024^2 decoder of configs[4] (4 images per GPU).

Prints ONE JSON line.  LPIPS / ArcFace-ID terms of the reference's training script need downloaded weights and are off."""
import argparse
import json
import time
import types

import torch

FP32_MFMA_PEAK = 157.3
BF16_MFMA_PEAK = 2516.6  # 16 x the fp32 matrix rate (MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, ~2.5 PFLOP/s dense)
HBM_PEAK_GBS = 8000.0


def synth(n, dev, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, 3, 256, 256, generator=g) - 0.5) / 0.5
    ref = (torch.rand(n, 3, 256, 256, generator=g) - 0.5) / 0.5
    y = (torch.rand(n, 3, 256, 256, generator=g) - 0.5) / 0.5
    m = torch.zeros(n, 256, 256)
    yy, xx = torch.meshgrid(torch.arange(256), torch.arange(256), indexing="ij")
    for i in range(n):
        cy, cx = 176 + int(torch.randint(-16, 17, (1,), generator=g)), 128 + int(torch.randint(-16, 17, (1,), generator=g))
        ry, rx = 56 + int(torch.randint(-12, 13, (1,), generator=g)), 80 + int(torch.randint(-12, 13, (1,), generator=g))
        m[i] = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).float()
    return x.to(dev), ref.to(dev), y.to(dev), m.to(dev)


def summarise(recs, peak=FP32_MFMA_PEAK, only_bf16=False):
    conv = [(t, f, s.elapsed_time(e)) for t, f, s, e in recs if not t.startswith("bytes:") and (not only_bf16 or "_bf16|" in t)]
    byt = [(t, f, s.elapsed_time(e)) for t, f, s, e in recs if t.startswith("bytes:")]
    out = {}
    if conv:
        fl, ms = sum(f for _, f, _ in conv), sum(m for _, _, m in conv)
        out["mfma"] = {"launches": len(conv), "algorithmic_tflop": round(fl / 1e12, 3), "kernel_ms": round(ms, 3),
                       "tflops": round(fl / ms / 1e9, 2), "peak_tflops": peak, "utilisation": round(fl / ms / 1e9 / peak, 4)}
        # the same sum over the launches that carry the work (>= 100 GFLOP each): a launch is bracketed by events in an EAGER step, so
        # the 4^2 .. 16^2 layers (30-60 us each, mostly the gap to the next launch) weigh on the figure above far beyond their FLOPs
        big = [(f, m) for _, f, m in conv if f >= 1e11]
        if big:
            fl, ms = sum(f for f, _ in big), sum(m for _, m in big)
            out["mfma"]["large_launches"] = {"launches": len(big), "algorithmic_tflop": round(fl / 1e12, 3), "kernel_ms": round(ms, 3),
                                             "tflops": round(fl / ms / 1e9, 2), "utilisation": round(fl / ms / 1e9 / peak, 4)}
    for key in ("upfirdn2d", "noise_bias_act"):
        sel = [(f, m) for t, f, m in byt if t.startswith("bytes:" + key)]
        if sel:
            b, ms = sum(f for f, _ in sel), sum(m for _, m in sel)
            out[key] = {"launches": len(sel), "algorithmic_GB": round(b / 1e9, 3), "kernel_ms": round(ms, 3), "GBps": round(b / ms / 1e6, 1),
                        "frac_of_hbm_peak": round(b / ms / 1e6 / HBM_PEAK_GBS, 4)}
            big = [(f, m) for f, m in sel if f >= 64e6]  # launches that move >= 64 MB
            if big:
                b, ms = sum(f for f, _ in big), sum(m for _, m in big)
                out[key]["large_launches"] = {"launches": len(big), "algorithmic_GB": round(b / 1e9, 3), "kernel_ms": round(ms, 3),
                                              "GBps": round(b / ms / 1e6, 1), "frac_of_hbm_peak": round(b / ms / 1e6 / HBM_PEAK_GBS, 4)}
    return out


def decoder_profile(size, n, dev, backward=True, dtype=torch.float32):
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd.modules.psp.stylegan2.model import Generator

    torch.manual_seed(0)
    gen = Generator(size, 512, 8, compute_dtype=dtype).to(dev)
    lat = torch.randn(n, gen.n_latent, 512, device=dev, requires_grad=True)

    def run():
        img, _ = gen([lat], input_is_latent=True, randomize_noise=True)
        if backward:
            img.square().mean().backward()
        return img

    run()
    torch.cuda.synchronize()
    FF.PROFILE = []
    run()
    torch.cuda.synchronize()
    recs, FF.PROFILE = FF.PROFILE, None
    del gen
    torch.cuda.empty_cache()
    if dtype == torch.bfloat16:  # ModulatedConv2d launches only (the EqualLinear style GEMMs stay fp32)
        return summarise(recs, BF16_MFMA_PEAK, only_bf16=True)
    return summarise(recs)


def native_op_bandwidth(dev):
    """the reference's two native ops through their drop-in python names, planar [N*C, H, W] layout, fp32 and bf16, at the Blur shape
    of the 1024^2 decoder (4 images x 32 channels, 1025^2 -> 1024^2) and a 512^2 x 64 activation for the fused bias+lrelu"""
    from face_mask_inpaint_amd.modules.psp.stylegan2.op.fused_act import fused_bias_act
    from face_mask_inpaint_amd.modules.psp.stylegan2.op.upfirdn2d import _native

    k = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k = (k[None, :] * k[:, None] / 64 * 4).to(dev)
    out = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = torch.randn(128, 1025, 1025, device=dev).to(dt)
        a = torch.randn(4, 64, 512, 512, device=dev).to(dt)
        b = torch.randn(64, device=dev).to(dt)
        empty = a.new_empty(0)
        res = {}
        for key, fn, nbytes in (("upfirdn2d", lambda: _native(x, k, 1, 1, 1, 1, 1, 1, 1, 1), (128 * 1025 * 1025 + 128 * 1024 * 1024) * x.element_size()),
                                ("fused_bias_act", lambda: fused_bias_act(a, b, empty, 3, 0, 0.2, 2 ** 0.5), 2 * a.numel() * a.element_size())):
            for _ in range(5):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(30):
                fn()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 30
            res[key] = {"ms": round(ms, 3), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
        out[name] = res
    return out


LOSS_ARGS = dict(id_lambda=0, lpips_lambda=0, l2_lambda=1.0, style_lambda=0, lpips_lambda_ref=0, l2_lambda_ref=1.0, cx_lambda=0, w_norm_lambda=0.005,
                 start_from_latent_avg=True)


# the loss of the reference's shipped training script (scripts/train_psp.sh:12-15; the *_ref lambdas keep train_psp.py's defaults of 0):
# ArcFace ID + masked L2 + masked LPIPS(alex) in the backward, VGG style / contextual evaluated for the log only
SCRIPT_LOSS_ARGS = dict(id_lambda=0.1, lpips_lambda=0.8, l2_lambda=2.0, style_lambda=1000.0, lpips_lambda_ref=0.0, l2_lambda_ref=0.0, cx_lambda=1.0,
                        w_norm_lambda=0.0, start_from_latent_avg=True)


def train_leg(dev, decoder_dtype="bf16", size=256, batch=16, steps=5, warmup=2, train_decoder=True, loss_args=None, graph=False, encoder_dtype="fp32"):
    """one train_psp.py step loop (train_psp.py:307-335): pSp forward (GradualStyleEncoder on src + ref with attention, StyleGAN2
    decoder of ``size``), pSpLoss, backward, fused Adam over the encoder (+ decoder when train_decoder, as scripts/train_psp.sh runs
    it).  Returns (seconds for ``steps`` steps, per-launch summary of one extra profiled step).
    ``graph=True``: the whole step (forward, loss, backward, optimiser) is captured ONCE in a HIP graph (torch.cuda.graph) and
    replayed -- ~5 500 launches per step are CPU-bound at 4 images per GPU; the captured work is identical (fresh N(0,1) noise per
    replay from the graph-safe generator, device-side Adam step count, loss values read back after the replay)."""
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from face_mask_inpaint_amd.optim import FusedAdam

    torch.manual_seed(0)
    opts = types.SimpleNamespace(output_size=size, encoder_type="GradualStyleEncoder", train_decoder=train_decoder, use_attention=True, pt_ckpt_path=None,
                                 stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype=decoder_dtype, encoder_dtype=encoder_dtype)
    net = pSp(opts).to(dev).train()
    net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
    crit = pSpLoss(types.SimpleNamespace(**(loss_args or LOSS_ARGS)))
    if hasattr(crit, "to"):
        crit = crit.to(dev)
    crit.defer_logs = graph
    params = [p for p in net.encoder.parameters() if p.requires_grad]
    if train_decoder:
        params += [p for p in net.decoder.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, capturable=graph)
    x, ref, y, m = synth(batch, dev)

    def step():
        y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True)
        loss, _, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
        opt.zero_grad()
        loss.backward()
        if graph:  # train_psp.py:328-331 skips a step whose loss is not finite; in the captured step that test runs on the device
            opt.step(guard=loss.detach().reshape(1))
        else:
            if torch.isfinite(loss):
                opt.step()
        return loss

    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager warm-up on a side stream (allocator pools, lazily created state), as capture requires
            for _ in range(max(warmup, 2)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_loss = step()
        eager_step, step = step, lambda: (g.replay(), static_loss)[1]
        step()
    else:
        for _ in range(warmup):
            step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).item()
    if graph:
        step = eager_step  # the profiled step below brackets every launch with events: eager
    FF.PROFILE = []
    step()
    torch.cuda.synchronize()
    recs, FF.PROFILE = FF.PROFILE, None
    del net, opt, crit
    torch.cuda.empty_cache()
    bf16 = decoder_dtype in ("bf16", torch.bfloat16)
    summ = summarise(recs, BF16_MFMA_PEAK, only_bf16=True) if bf16 else summarise(recs)
    return dt, summ


def extra_block(dev, steps=4, warmup=2):
    """what bench.py appends to its JSON line so that the driver's fixed command also times BASELINE configs[2] (C3) and the single-GPU
    leg of configs[4] (C5): the train_psp step with the bf16 decoder -- images/s, ModulatedConv2d bf16 TFLOP/s inside that step (all
    bf16 convolution launches of one step: forward, adjoint, weight gradient) against the 2516.6 TFLOP/s dense bf16 peak, the
    in-decoder upfirdn2d (Blur / RGB-skip Upsample, forward + backward) and fused noise+bias+lrelu in GB/s against 8 TB/s"""
    out = {}
    for key, size, batch in (("C3_train_psp_256_bf16_bs16", 256, 16), ("C5_train_psp_1024_bf16_bs4_single_gpu_leg", 1024, 4)):
        dt, summ = train_leg(dev, "bf16", size, batch, steps, warmup, train_decoder=True, loss_args=SCRIPT_LOSS_ARGS, graph=True, encoder_dtype="bf16")
        dt32, summ32 = train_leg(dev, "bf16", size, batch, steps, warmup, train_decoder=True, loss_args=SCRIPT_LOSS_ARGS, graph=True, encoder_dtype="fp32")
        mf, mfa = summ32.get("mfma", {}), summ.get("mfma", {})  # with the fp32 body every bf16 convolution launch is a ModulatedConv2d
        # headline = the configuration as BASELINE.json states it (bf16 in the StyleGAN2 DECODER only, fp32 IR-SE50 body); the variant with
        # the encoder body in bf16 as well is reported beside it
        out[key] = {"images_per_s": round(batch * steps / dt32, 2), "ms_per_step": round(dt32 / steps * 1e3, 2),
                    "bf16_encoder_body_variant": {"images_per_s": round(batch * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 2)},
                    "modulated_conv_bf16": {"tflops": mf.get("tflops"), "frac_of_bf16_peak": mf.get("utilisation"), "launches": mf.get("launches"),
                                            "kernel_ms": mf.get("kernel_ms"), "algorithmic_tflop": mf.get("algorithmic_tflop"),
                                            "large_launches_ge_100_gflop": mf.get("large_launches")},
                    "all_conv_bf16_incl_encoder_body": {"tflops": mfa.get("tflops"), "frac_of_bf16_peak": mfa.get("utilisation"), "launches": mfa.get("launches"),
                                                        "kernel_ms": mfa.get("kernel_ms"), "algorithmic_tflop": mfa.get("algorithmic_tflop")},
                    "upfirdn2d_in_decoder": summ.get("upfirdn2d"), "noise_bias_act": summ.get("noise_bias_act"),
                    "config": "train_psp.py RefpSp + attention, fp32 IR-SE50 body (src + ref as one 2N batch), StyleGAN2 %d^2 decoder with bf16 activations (fp32 accumulate / statistics / "
                              "master weights; style heads, attention and the loss networks fp32), bs %d, --train_decoder 1 and the loss of scripts/train_psp.sh (ArcFace ID 0.1 + "
                              "masked L2 2 + masked LPIPS-alex 0.8 in the backward; VGG style / contextual logged only; random-init LPIPS / ArcFace weights), fused Adam; the whole step captured "
                              "once in a HIP graph and replayed; %d timed steps; a non-finite loss turns the optimiser step into a device-side no-op (train_psp.py:328-331).  bf16_encoder_body_variant: the same "
                              "step with opts.encoder_dtype=bf16 (the 24 IR-SE bottlenecks keep bf16 activations as well: narrower than the config, reported beside it); "
                              "modulated_conv_bf16 is measured in the headline step, where every bf16 convolution launch is a ModulatedConv2d; upfirdn2d / noise_bias_act in the variant step" % (size, batch, steps)}
    out["peaks"] = {"bf16_mfma_tflops": BF16_MFMA_PEAK, "hbm_GBps": HBM_PEAK_GBS}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--skip-1024", action="store_true")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    dev = torch.device("cuda:0")
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from face_mask_inpaint_amd.optim import FusedAdam

    dt, whole = train_leg(dev, "fp32", 256, args.batch, args.steps, args.warmup, train_decoder=False)
    dt16, whole16 = train_leg(dev, "bf16", 256, args.batch, args.steps, args.warmup, train_decoder=False)
    out = {"metric": "train_psp images/sec (fp32, encoder trained, decoder frozen, LPIPS/ID off)", "value": round(args.batch * args.steps / dt, 2), "unit": "images/s",
           "n_gpus": 1, "batch": args.batch, "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 2), "dtype": "f32", "data": "synthetic",
           "whole_step": whole,
           "bf16_decoder": {"config": "BASELINE.json configs[2]: StyleGAN2 256^2 decoder in bf16 (fp32 accumulate), encoder fp32, bs %d" % args.batch,
                            "images_per_s": round(args.batch * args.steps / dt16, 2), "ms_per_step": round(dt16 / args.steps * 1e3, 2),
                            "whole_step": whole16,
                            "decoder_256_fwd_bwd": decoder_profile(256, args.batch, dev, dtype=torch.bfloat16)},
           "decoder_256_fwd_bwd": decoder_profile(256, args.batch, dev),
           "native_ops_planar": native_op_bandwidth(dev),
           "peaks": {"fp32_mfma_tflops": FP32_MFMA_PEAK, "bf16_mfma_tflops": BF16_MFMA_PEAK, "hbm_GBps": HBM_PEAK_GBS}}
    if not args.skip_1024:
        out["decoder_1024_fwd_bs4"] = decoder_profile(1024, 4, dev, backward=False)
        out["decoder_1024_fwd_bwd_bs4"] = decoder_profile(1024, 4, dev, backward=True)
        out["bf16_decoder"]["decoder_1024_fwd_bwd_bs4"] = decoder_profile(1024, 4, dev, backward=True, dtype=torch.bfloat16)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
