"""GPU: the device side of the data path (csrc/preproc.hip, face_mask_inpaint_amd/preprocess.py, dataloader.py) against the tensors the
reference's own ReferenceDataset returned for the committed files (tests/golden/dataset.pt) -- integer / table work, so bit exact --
and the pytorch_msssim-style SSIM / MS-SSIM metric against its CPU restatement (parity unpinned: the package is absent, no reference
fixture exists; tolerance 2e-5 = fp32 filtering noise)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "dataset")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


def _dataset(transform, **kw):
    from face_mask_inpaint_amd.dataloader import ReferenceDataset

    return ReferenceDataset(os.path.join(DATA, "images_masked"), os.path.join(DATA, "images"), os.path.join(DATA, "binary_map"),
                            os.path.join(DATA, "identity.txt"), apply_transform=transform, scale=0.5, return_id=True, **kw)


@pytest.mark.parametrize("transform", [False, True])
def test_reference_dataset_items_equal_the_reference(dev, golden, transform):
    """every item of the tiny data set, produced by the GPU kernels, equals the reference's CPU pipeline bit for bit (same dtypes, same
    sampled reference image under the fixture's seeds)"""
    fx = golden("dataset.pt")["transform" if transform else "plain"]
    ds = _dataset(transform, device=dev)
    assert sorted(ds.ids) == fx["ids"]
    for want_id, want in zip(fx["ids"], fx["items"]):
        random.seed(1000 + int(want_id))
        got = ds[ds.ids.index(want_id)]
        assert set(got) == set(want)
        for k in want:
            assert got[k].is_cuda and got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, (want_id, k)
            assert torch.equal(got[k].cpu(), want[k]), (want_id, k)


def test_device_loader_batches_and_training_boundary(dev, golden):
    """a batch from DeviceLoader = the stacked items (files decoded on the thread pool, one preprocessing launch per tensor kind), and
    to_device_batch adds the bit-exact (mask > 0).float() of train_reference_fill.py:340"""
    from face_mask_inpaint_amd.dataloader import DeviceLoader, to_device_batch

    fx = golden("dataset.pt")["plain"]
    ds = _dataset(False, device=dev)
    order = [ds.ids.index(i) for i in fx["ids"]]
    seen = 0
    # the loader draws all reference partners of a batch up front, on the calling thread: replay the same draws item by item
    loader = DeviceLoader(ds, order, batch_size=3, shuffle=False, drop_last=False, num_workers=2)
    assert len(loader) == 3
    random.seed(7)
    batches = list(loader)
    random.seed(7)
    for b in batches:
        n = b["mask"].shape[0]
        for j in range(n):
            key = fx["ids"][seen + j]
            partner = ds.sample_reference_image(key)
            want = fx["items"][seen + j]
            assert torch.equal(b["src_img"][j].cpu(), want["src_img"]) and torch.equal(b["mask"][j].cpu(), want["mask"])
            assert torch.equal(b["gt_img"][j].cpu(), want["gt_img"]) and int(b["id"][j]) == int(key)
            ref_want = [it for i, it in zip(fx["ids"], fx["items"]) if i == partner][0]["raw_gt_img"]  # the partner's own ground truth
            assert torch.equal(b["ref_img"][j].cpu(), ref_want)
        seen += n
    assert seen == 7
    out = to_device_batch(batches[0])
    assert torch.equal(out["true_masks"].cpu(), (batches[0]["mask"].cpu() > 0).float())


@pytest.mark.parametrize("h,w,scale", [(96, 80, 0.25), (67, 131, 0.6), (128, 128, 1.0), (51, 49, 0.9)])
def test_preprocessor_against_the_oracle(dev, h, w, scale):
    """random 8-bit images / masks at sizes with ragged tables: both resampling passes, the NEAREST gather and the table cast against
    the oracle's restatement of Pillow (itself checked against Pillow on the CPU)"""
    from face_mask_inpaint_amd.preprocess import DevicePreprocessor
    from oracle import pil_resize_cpu as O

    rng = np.random.RandomState(h * 7 + w)
    pre = DevicePreprocessor(dev)
    imgs = [rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(3)]
    imgs[1][:] = 255
    imgs[2][::2] = 0  # hard edges: the cubic overshoots below 0 and above 255 (clip8)
    masks = [(rng.randint(0, 2, (h, w)) * 255).astype(np.uint8) for _ in range(3)]
    got = pre.images(imgs, scale).cpu()
    gotn, gotp = pre.images(imgs, scale, normalise=True, also_plain=True)
    gm = pre.masks(masks, scale).cpu()
    for i in range(3):
        want = torch.from_numpy(O.preprocess(imgs[i], scale, False))
        assert torch.equal(got[i], want) and torch.equal(gotp[i].cpu(), want) and torch.equal(gotn[i].cpu(), (want - 0.5) / 0.5)
        assert torch.equal(gm[i], torch.from_numpy(O.preprocess(masks[i], scale, True)))
    assert gm.dtype == torch.int64 and got.dtype == torch.float32


def test_find_best_reference_on_the_ssim_kernel(dev, tmp_path):
    """use_ssim: for every image the same-identity partner of highest SSIM, scored on the GPU; the choice equals an argmax over the CPU
    oracle's SSIM of the same preprocessed images, and the JSON cache round-trips"""
    import shutil

    from face_mask_inpaint_amd.dataloader import ReferenceDataset, decode
    from oracle import pil_resize_cpu as O
    from oracle import ssim_cpu

    root = tmp_path / "data"
    shutil.copytree(DATA, root)
    ds = ReferenceDataset(str(root / "images_masked"), str(root / "images"), str(root / "binary_map"), str(root / "identity.txt"),
                          apply_transform=False, scale=0.5, use_ssim=True, device=dev)
    assert (root / "best_reference_map.json").is_file()
    f = lambda k: torch.from_numpy(O.preprocess(decode(str(root / "images" / (k + ".jpg"))), 0.5, False)).unsqueeze(0)
    for key in ds.ids:
        cands = ds.partners(key)
        scores = [float(ssim_cpu.ssim(f(key), f(c))) for c in cands]
        assert ds.best_reference_map[key] == cands[int(np.argmax(scores))], (key, scores)
        assert ds.sample_reference_image(key) == ds.best_reference_map[key]
    again = ReferenceDataset(str(root / "images_masked"), str(root / "images"), str(root / "binary_map"), str(root / "identity.txt"),
                             apply_transform=False, scale=0.5, use_ssim=True, device=dev)
    assert again.best_reference_map == ds.best_reference_map


@pytest.mark.parametrize("shape", [(2, 3, 192, 180), (1, 3, 256, 256), (1, 1, 177, 201)])
def test_ms_ssim_against_the_cpu_restatement(dev, shape):
    """SSIM / MS-SSIM of the trainers' metric package (valid Gaussian filtering, five scales): HIP kernels against oracle/msssim_cpu.py in
    float64 -- 2e-5 absolute; identical images give exactly 1; two runs are bit-identical (fixed-order reduction)"""
    from face_mask_inpaint_amd.modules.evaluations.msssim import MS_SSIM, SSIM
    from oracle import msssim_cpu as M

    g = torch.Generator().manual_seed(shape[2])
    x = torch.rand(shape, generator=g)
    y = (x + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
    xd, yd = x.to(dev), y.to(dev)
    s_fn, m_fn = SSIM(data_range=1, size_average=True, channel=shape[1]), MS_SSIM(data_range=1, size_average=True, channel=shape[1])
    s, m = s_fn(xd, yd), m_fn(xd, yd)
    assert abs(float(s) - float(M.ssim(x.double(), y.double()))) <= 2e-5
    assert abs(float(m) - float(M.ms_ssim(x.double(), y.double()))) <= 2e-5
    assert float(m_fn(xd, xd)) == 1.0 and float(s_fn(xd, xd)) == 1.0
    assert float(m_fn(xd, yd)) == float(m) and float(s_fn(xd, yd)) == float(s)
    per = MS_SSIM(data_range=1, size_average=False)(xd, yd)
    torch.testing.assert_close(per.cpu().double(), M.ms_ssim(x.double(), y.double(), size_average=False), rtol=0, atol=2e-5)
    with pytest.raises(AssertionError):
        m_fn(xd[..., :150, :150], yd[..., :150, :150])
