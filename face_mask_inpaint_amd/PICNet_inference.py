"""Host-side mirror of the reference's inference harness, PICNet_inference.py (BASELINE.json configs[0]): ``process_params``
(:73-85), ``infer_batch`` (:88-109), ``tensor2im`` (:112-117), ``evaluate`` (:120-124) and a ``main`` that runs them on synthetic
CelebA-HQ-shaped batches or on a ReferenceDataset directory.  The generator and the mask detector run on the HIP kernels; there is
no CPU path (the CPU counterpart used for parity is oracle/unet_cpu.py:infer_batch).

SSIM: the reference imports pytorch_msssim (absent offline, parity unpinned -- SURVEY.md 8c); ``evaluate`` here takes any callable
and ``main`` uses this build's SSIM kernel (modules/evaluations/ssim.py definition: 11-tap Gaussian, zero padded)."""
from __future__ import annotations

import argparse
import os

import torch

from . import functional as FF
from .modules.mask_detector import MaskDetector
from .modules.model import ReferenceFill, scale_img
from .modules.pluralistic_model import base_function


def get_args(argv=None):
    """the reference's flags for the model side (PICNet_inference.py:18-70); data flags default to synthetic input"""
    p = argparse.ArgumentParser()
    p.add_argument("--data_root", type=str, default=None, help="directory with src / ref / mask sub-directories; synthetic batches when absent")
    p.add_argument("--src_img_path", type=str, default="images_masked")
    p.add_argument("--ref_img_path", type=str, default="images")
    p.add_argument("--mask_path", type=str, default="binary_map")
    p.add_argument("--identity_file_path", type=str, default="CelebA-HQ-identity.txt")
    p.add_argument("--mask_detector_path", type=str, default=None)
    p.add_argument("--pt_ckpt_path", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--img_scale", type=float, default=0.25)
    p.add_argument("--use_att", type=int, default=1)
    p.add_argument("--old_model", action="store_true")
    p.add_argument("--use_best_reference", action="store_true")
    p.add_argument("--save_src_mask", action="store_true")
    p.add_argument("--num_batches", type=int, default=2, help="synthetic mode: batches to run")
    p.add_argument("--encoder_type", type=str, default="pluralistic")
    p.add_argument("--encoder_ngf", type=int, default=32)
    p.add_argument("--encoder_z_nc", type=int, default=128)
    p.add_argument("--encoder_img_f", type=int, default=128)
    p.add_argument("--encoder_layers", type=int, default=5)
    p.add_argument("--encoder_norm", type=str, default="none")
    p.add_argument("--encoder_activation", type=str, default="LeakyReLU")
    p.add_argument("--encoder_L", type=int, default=6)
    p.add_argument("--decoder_ngf", type=int, default=32)
    p.add_argument("--decoder_z_nc", type=int, default=256)
    p.add_argument("--decoder_img_f", type=int, default=256)
    p.add_argument("--decoder_L", type=int, default=0)
    p.add_argument("--decoder_layers", type=int, default=5)
    p.add_argument("--decoder_norm", type=str, default="instance")
    p.add_argument("--decoder_activation", type=str, default="LeakyReLU")
    return p.parse_args(argv)


def process_params(args):
    encoder_params = {k.replace("encoder_", ""): v for k, v in args._get_kwargs() if k.startswith("encoder")}
    decoder_params = {k.replace("decoder_", ""): v for k, v in args._get_kwargs() if k.startswith("decoder")}
    return encoder_params, decoder_params


@torch.no_grad()
def infer_batch(generator, mask_detector, batch_images, device, old_model=False, eps=None):
    """PICNet_inference.py:88-109.  ``eps`` (this build's extra) injects the two rsample draws for reproducible comparisons."""
    generator.eval()
    if len(batch_images) == 1:
        src_img = batch_images[0].to(device)
        ref_img = src_mask = None
    else:
        src_img, ref_img = batch_images
        src_img = src_img.to(device)
        ref_img = ref_img.to(device)
        if hasattr(mask_detector, "predict_mask"):
            src_mask = mask_detector.predict_mask(src_img)                      # argmax as one index kernel, bit exact
        else:
            src_mask = mask_detector(src_img, mode="train").argmax(1).float()   # [N, H, W]
    if old_model:
        src_img = scale_img(src_img, (218, 178))
        ref_img = scale_img(ref_img, (218, 178))
    gen_images = generator(src_img, ref_img, src_mask=src_mask, no_prior=old_model, eps=eps)
    return gen_images.detach(), src_mask.detach().cpu()


def tensor2im(var):
    from PIL import Image

    var = var.permute(1, 2, 0).numpy().copy()
    var[var < 0] = 0
    var[var > 1] = 1
    return Image.fromarray((var * 255).astype("uint8"))


def evaluate(gt_img, gen_img, ssim_func, ms_ssim_func=None):
    ssim = ssim_func(gt_img, gen_img)
    ms = ms_ssim_func(gt_img, gen_img) if ms_ssim_func is not None else float("nan")
    return float(ssim), float(ms)


def build(args, device):
    mask_detector = MaskDetector(n_channels=3, bilinear=True)
    if args.mask_detector_path:
        mask_detector.load_state_dict(torch.load(args.mask_detector_path, map_location="cpu", weights_only=True))
    base_function._freeze(mask_detector)
    mask_detector = mask_detector.to(device).eval()
    encoder_params, decoder_params = process_params(args)
    kw = dict(out_size=(218, 178)) if args.old_model else {}
    generator = ReferenceFill(None, encoder_params, decoder_params, use_att=bool(args.use_att), **kw).to(device)
    if args.pt_ckpt_path:
        generator.load_state_dict(torch.load(args.pt_ckpt_path, map_location="cpu", weights_only=True), strict=False)
    return generator.eval(), mask_detector


def main(argv=None):
    from .modules.evaluations.msssim import MS_SSIM
    from .modules.evaluations.ssim import ssim as ssim_func

    ms_ssim_func = MS_SSIM(data_range=1, size_average=True, channel=3)  # PICNet_inference.py:130-131 in the reference
    args = get_args(argv)
    if not torch.cuda.is_available():
        raise FF.FmiError("PICNet_inference needs the MI355X (the HIP path has no CPU fallback)")
    device = torch.device("cuda:0")
    generator, mask_detector = build(args, device)
    if args.data_root:
        from .dataloader import DeviceLoader, ReferenceDataset

        j = lambda p: os.path.join(args.data_root, p)
        ds = ReferenceDataset(j(args.src_img_path), j(args.ref_img_path), j(args.mask_path), j(args.identity_file_path), apply_transform=False,
                              scale=args.img_scale, use_ssim=args.use_best_reference, device=device, return_id=True)
        batches = DeviceLoader(ds, range(len(ds)), args.batch_size, shuffle=False, drop_last=False)  # files decoded on the host, pixels made on the GPU
    else:
        size = int(1024 * args.img_scale)
        g = torch.Generator().manual_seed(0)
        batches = [{"src_img": torch.rand(args.batch_size, 3, size, size, generator=g), "ref_img": torch.rand(args.batch_size, 3, size, size, generator=g),
                    "raw_gt_img": torch.rand(args.batch_size, 3, size, size, generator=g)} for _ in range(args.num_batches)]
    results = []
    for batch in batches:
        gen_images, src_mask = infer_batch(generator, mask_detector, (batch["src_img"], batch["ref_img"]), device, args.old_model)
        gt = batch["raw_gt_img"].to(device)
        if args.old_model:
            gt = scale_img(gt, (218, 178))
        ms = ms_ssim_func if min(gt.shape[2:]) > 160 else None  # five scales need more than (11 - 1) * 2^4 pixels per side
        results.append(evaluate(gt.contiguous(), gen_images.contiguous().clamp(0, 1), ssim_func, ms))
    mean_ssim = sum(r[0] for r in results) / max(len(results), 1)
    mean_ms = sum(r[1] for r in results) / max(len(results), 1)
    print({"ssim": mean_ssim, "ms_ssim": mean_ms, "batches": len(results), "image": tuple(gen_images.shape)})
    return mean_ssim


if __name__ == "__main__":
    main()
