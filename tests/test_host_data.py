"""CPU: data and checkpoint formats either side of the hot path: ReferenceDataset against the reference's own
class on the committed tiny dataset, load_networks / get_keys / load_weights round trips on reference-keyed state_dicts."""
import os
import random
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "dataset")


@pytest.mark.parametrize("transform", [False, True])
def test_reference_dataset_host_side_and_oracle(golden, transform):
    """tests/golden/dataset/* are data files (jpg / npy / identity list); tests/golden/dataset.pt holds what the reference's
    ReferenceDataset returned for them (oracle/gen_golden.py:dataset_fixture).  Without a GPU this checks (a) the HOST side of the
    rewritten data path -- same ids (the singleton identity filtered), identity groups, the reference image drawn under the fixture's
    seeds, file decoding -- and (b) the oracle's restatement of Pillow's BICUBIC / NEAREST resizing + / 255 (+ Normalize) against the
    reference's tensors, bit for bit.  The device kernels face the same fixture in tests/test_gpu_data.py."""
    import numpy as np

    from face_mask_inpaint_amd import dataloader as DL
    from oracle import pil_resize_cpu as O

    fx = golden("dataset.pt")["transform" if transform else "plain"]
    ds = DL.ReferenceDataset(os.path.join(DATA, "images_masked"), os.path.join(DATA, "images"), os.path.join(DATA, "binary_map"),
                             os.path.join(DATA, "identity.txt"), apply_transform=transform, scale=0.5, return_id=True)
    assert sorted(ds.ids) == fx["ids"] and "108" not in ds.ids and ds.filter_id == {"108"}
    assert ds.identity_map[2] == ["103", "104", "105"] and ds.img2identity["107"] == 3 and ds.partners("104") == ["103", "105"]
    norm = (lambda t: (t - 0.5) / 0.5) if transform else (lambda t: t)
    for want_id, want in zip(fx["ids"], fx["items"]):
        random.seed(1000 + int(want_id))
        partner = ds.sample_reference_image(want_id)
        assert partner != want_id and partner in ds.partners(want_id)
        src, gt, ref, m = ds._load(want_id, partner)
        assert src.dtype == np.uint8 and src.shape == (48, 40, 3) and m.shape == (48, 40)
        f = lambda a: torch.from_numpy(O.preprocess(a, 0.5, False))
        assert torch.equal(norm(f(src)), want["src_img"]) and torch.equal(f(gt), want["raw_gt_img"])
        assert torch.equal(norm(f(gt)), want["gt_img"]) and torch.equal(norm(f(ref)), want["ref_img"]), (want_id, partner)
        assert torch.equal(torch.from_numpy(O.preprocess(m, 0.5, True)), want["mask"])
    with pytest.raises(Exception):  # pixels are made on the GPU: no CPU preprocessing path
        if not torch.cuda.is_available():
            ds[0]
        else:
            raise RuntimeError("skip")


def test_resampling_tables_and_oracle_against_pillow():
    """the host-side tables of face_mask_inpaint_amd.preprocess (vectorised) equal the oracle's loop restatement of Pillow's
    precompute_coeffs / ImagingScaleAffine, and the oracle equals Pillow itself where Pillow is importable"""
    import numpy as np

    from face_mask_inpaint_amd import preprocess as P
    from oracle import pil_resize_cpu as O

    for i, o in [(48, 24), (40, 20), (1024, 256), (53, 37), (100, 33), (50, 45), (31, 31), (7, 3), (218, 54)]:
        b, k, ks = P.bicubic_tables(i, o)
        ob, ok, oks = O.precompute_coeffs(i, o)
        assert ks == oks and np.array_equal(b, ob) and np.array_equal(k, ok), (i, o)
        assert np.array_equal(P.nearest_table(i, o), O.nearest_table(i, o))
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(0)
    for h, w, s in [(48, 40, 0.5), (64, 64, 0.25), (37, 53, 0.7), (100, 80, 0.33), (31, 29, 1.0)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        nw, nh = int(s * w), int(s * h)
        assert np.array_equal(np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BICUBIC)), O.resize_bicubic_u8(img, nw, nh))
        m = (rng.randint(0, 2, (h, w)) * 255).astype(np.uint8)
        assert np.array_equal(np.asarray(Image.fromarray(m).resize((nw, nh), resample=Image.NEAREST)), O.resize_nearest(m, nw, nh))


def test_device_loader_partitions_the_dataset():
    """get_reference_dataloader: a random split into train / validation index sets that cover the data set once; batch counts follow
    drop_last (host logic only: iterating needs the GPU)"""
    from face_mask_inpaint_amd import dataloader as DL

    torch.manual_seed(0)
    tr, va = DL.get_reference_dataloader(os.path.join(DATA, "images_masked"), os.path.join(DATA, "images"), os.path.join(DATA, "binary_map"),
                                         os.path.join(DATA, "identity.txt"), batch_size=2, val_amount=0.3, img_scale=0.5)
    assert sorted(tr.indices + va.indices) == list(range(7)) and len(va.indices) == 3 and len(tr) == 2 and len(va) == 1
    assert tr.shuffle and not va.shuffle and va.drop_last and tr.dataset is va.dataset


def _tiny_models():
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from face_mask_inpaint_amd.modules.pluralistic_model import network

    enc = dict(type="pluralistic", ngf=8, z_nc=8, img_f=16, layers=5, norm="none", activation="LeakyReLU", L=2)
    dec = dict(ngf=8, z_nc=16, img_f=32, layers=5, norm="instance", activation="LeakyReLU", L=0)
    G = ReferenceFill(None, dict(enc), dict(dec), use_att=True, out_size=(64, 64))
    D = network.define_d(ndf=8, img_f=32, layers=4, norm="none", activation="LeakyReLU", model_type="ResDis")
    return G, D


def test_load_networks_round_trip(tmp_path):
    """train_reference_fill.py:107-140 on PICNet-style checkpoints saved under DataParallel (keys prefixed ``module.``): D is loaded
    strictly; G / E keep the model's own values for shape-matching keys (the reference's ``matches[k] = v`` takes v from the model),
    copy_pretrained=True really copies; a mismatching shape is skipped; a key missing from the file raises KeyError as there"""
    from face_mask_inpaint_amd.train_reference_fill import load_networks, process_params

    torch.manual_seed(0)
    G0, D0 = _tiny_models()
    torch.manual_seed(1)
    G, D = _tiny_models()
    dp = lambda sd: {"module." + k: v.clone() for k, v in sd.items()}
    e_sd = dp(G0.src_encoder.state_dict())
    # a PICNet encoder checkpoint carries BOTH heads; here the reference encoder's posterior block rides along, and one tensor has a
    # foreign shape (must be skipped, not copied)
    for k, v in G0.ref_encoder.state_dict().items():
        e_sd.setdefault("module." + k, v.clone())
    src_sd = G0.src_encoder.state_dict()
    ptr = src_sd["encoder0.conv1.module.bias"].data_ptr()
    odd = [k for k, v in src_sd.items() if v.data_ptr() == ptr]  # conv1 and its nn.Sequential alias model.N share one tensor
    assert len(odd) == 2
    for k in odd:
        e_sd["module." + k] = torch.zeros(src_sd[k].numel() + 3)
    torch.save(dp(G0.decoder.state_dict()), tmp_path / "latest_net_G.pth")
    torch.save(e_sd, tmp_path / "latest_net_E.pth")
    torch.save(dp(D0.state_dict()), tmp_path / "latest_net_D.pth")
    before = {k: v.clone() for k, v in G.state_dict().items()}
    load_networks(G, D, str(tmp_path))
    for k, v in D.state_dict().items():
        assert torch.equal(v, D0.state_dict()[k]), k
    for k, v in G.state_dict().items():
        assert torch.equal(v, before[k]), k  # the reference's literal behaviour: nothing from the G / E files arrives
    load_networks(G, D, str(tmp_path), copy_pretrained=True)
    for k, v in G.decoder.state_dict().items():
        assert torch.equal(v, G0.decoder.state_dict()[k]), k
    for k, v in G.src_encoder.state_dict().items():
        want = before["src_encoder." + k] if k in odd else G0.src_encoder.state_dict()[k]
        assert torch.equal(v, want), k
    load_networks(G, D, "")  # falsy path: no-op
    os.remove(tmp_path / "latest_net_G.pth")
    torch.save({k: v for k, v in dp(G0.decoder.state_dict()).items() if "out4" not in k}, tmp_path / "latest_net_G.pth")
    with pytest.raises(KeyError):
        load_networks(G, D, str(tmp_path))
    args = types.SimpleNamespace(encoder_ngf=8, encoder_img_f=16, decoder_ngf=8, decoder_z_nc=16, disc_ndf=8, disc_layers=4, other=1)
    args._get_kwargs = lambda: sorted(vars(args).items())
    e, d, c = process_params(args)
    assert e == {"ngf": 8, "img_f": 16} and d == {"ngf": 8, "z_nc": 16} and c == {"ndf": 8, "layers": 4, "img_f": 16}


def test_psp_checkpoint_round_trip(tmp_path):
    """psp.py:14-18,50-56: a pSp training checkpoint ({'state_dict': {'encoder.*', 'decoder.*'}, 'latent_avg'}) written from one
    instance loads into a second through opts.pt_ckpt_path: get_keys strips the prefixes, the decoder loads strictly, latent_avg
    arrives un-repeated"""
    from face_mask_inpaint_amd.modules.psp.psp import get_keys, pSp

    def opts(path=None):
        return types.SimpleNamespace(output_size=64, encoder_type="GradualStyleEncoder", use_attention=True, train_decoder=False,
                                     start_from_latent_avg=True, learn_in_w=False, pt_ckpt_path=path, stylegan_weights=None)

    torch.manual_seed(3)
    a = pSp(opts())
    lat = torch.randn(a.opts.n_styles, 512)
    sd = {"state_dict": {k: v for k, v in a.state_dict().items()}, "latent_avg": lat, "opts": {"output_size": 64}}
    assert set(get_keys(sd, "decoder")) == set(a.decoder.state_dict()) and set(get_keys(sd, "encoder")) == set(a.encoder.state_dict())
    path = str(tmp_path / "psp.pt")
    torch.save(sd, path)
    torch.manual_seed(4)
    b = pSp(opts(path))
    for k, v in b.state_dict().items():
        assert torch.equal(v, a.state_dict()[k]), k
    assert torch.equal(b.latent_avg, lat)
