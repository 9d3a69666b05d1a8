"""CPU: host logic of the drop-in layer -- state_dict compatibility with the reference (via the golden
fixtures), the C ABI export list, and the 'no CPU fallback' rule.  No kernel is launched."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from face_mask_inpaint_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "fmi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fmi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) > 40
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libfmi_hip.so not built (run __graft_entry__.build())")
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(cdll, s)]
    assert not missing, missing
    # the python binding covers every entry point except the two informational ones
    unbound = [s for s in declared if s not in _lib.SIGNATURES and s not in _lib.PREDICATES and s not in ("fmi_status_string", "fmi_version")]
    assert not unbound, unbound
    cdll.fmi_status_string.restype = ctypes.c_char_p
    assert cdll.fmi_status_string(2) == b"unsupported shape or mode"


def test_state_dict_keys_match_reference(golden):
    from face_mask_inpaint_amd.modules.loss import VGGLoss
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from face_mask_inpaint_amd.modules.pluralistic_model import network

    fx = golden("picnet_train_tiny.pt")
    cfg = fx["config"]
    enc = dict(type="pluralistic", ngf=8, z_nc=cfg["enc_z_nc"], img_f=16, layers=5, norm="none", activation="LeakyReLU", L=cfg["enc_L"])
    dec = dict(ngf=8, z_nc=16, img_f=32, layers=5, norm="instance", activation="LeakyReLU", L=0)
    G = ReferenceFill(None, dict(enc), dict(dec), use_att=True, out_size=(64, 64))
    D = network.define_d(ndf=8, img_f=32, layers=cfg["disc_layers"], norm="none", activation="LeakyReLU", model_type="ResDis")
    V = VGGLoss(width_div=cfg["vgg_div"])
    for mod, ref in ((G, fx["G_sd0"]), (D, fx["D_sd0"]), (V, fx["V_sd"])):
        sd = mod.state_dict()
        for k, v in ref.items():
            assert k in sd, k
            assert tuple(sd[k].shape) == tuple(v.shape), k
        extra = [k for k in sd if k not in ref]
        # the fixture drops the aliases model.N.module.* / shortcut.* of shared convs; nothing else may differ
        assert all(re.search(r"(^|\.)(shortcut\.|model\.\d+\.module\.)", k) for k in extra), extra
    # parameters that the reference trains are trainable here too, u/v are not
    for n, p in G.named_parameters():
        assert p.requires_grad == (not (n.endswith("weight_u") or n.endswith("weight_v"))), n
    # Auto_Attn.model is never executed (pre is None): excluded from weight preparation
    from face_mask_inpaint_amd.weights import _collect

    convs = _collect(G)
    inner = {id(m) for m in G.decoder.attn1.model.modules()}
    assert not any(id(c) in inner for c in convs)
    assert len(convs) == len({id(c) for c in convs})


def test_no_cpu_fallback():
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd._lib import FmiError, Library

    with pytest.raises(FmiError):
        FF.leaky_relu(torch.zeros(1, 2, 2, 4), 0.1)  # CPU tensor: refused, not computed
    with pytest.raises(FmiError):
        Library("/nonexistent/libfmi_hip.so")


def test_initialisation_follows_reference_rule():
    """SpectralNorm-wrapped convs keep the default init; plain convs under define_* get orthogonal(gain .02), zero bias."""
    from face_mask_inpaint_amd.modules.pluralistic_model import network

    torch.manual_seed(0)
    g = network.define_g(ngf=8, z_nc=16, img_f=32, layers=5, norm="instance", activation="LeakyReLU", L=0)
    q = g.attn1.query_conv
    assert torch.all(q.bias == 0)
    w = q.weight.view(q.weight.shape[0], -1)
    torch.testing.assert_close(w @ w.t(), 0.02 ** 2 * torch.eye(w.shape[0]), rtol=1e-4, atol=1e-7)
    assert float(g.attn1.gamma) == 0.0 and float(g.attn1.alpha) == 0.0
    u = g.decoder0.conv1.module.weight_u
    torch.testing.assert_close(u.norm(), torch.tensor(1.0), rtol=1e-5, atol=1e-6)


def test_psp_state_dict_keys_match_reference(golden):
    """the pSp-side mirrors take the reference's state_dicts unchanged (strict load), and pSp builds offline"""
    import types

    from face_mask_inpaint_amd.modules.psp.encoders import helpers as H
    from face_mask_inpaint_amd.modules.psp.encoders.psp_encoders import GradualStyleBlock, GradualStyleEncoder
    from face_mask_inpaint_amd.modules.psp.psp import get_keys, pSp

    fx = golden("psp_ops.pt")
    H.bottleneck_IR_SE(16, 32, 2).load_state_dict(fx["ir_se_conv_s2"]["sd"], strict=True)
    H.bottleneck_IR_SE(32, 32, 2).load_state_dict(fx["ir_se_pool_s2"]["sd"], strict=True)
    H.bottleneck_IR(8, 24, 2).load_state_dict(fx["ir_conv_s2"]["sd"], strict=True)
    GradualStyleBlock(16, 16, 8).load_state_dict(fx["style_block"]["sd"], strict=True)
    e = fx["encoder"]
    enc = GradualStyleEncoder(50, "ir_se", types.SimpleNamespace(n_styles=e["n_styles"], use_attention=True), _widths=tuple(e["widths"]),
                              _spatial=tuple(e["spatial"]))
    enc.load_state_dict(e["sd"], strict=True)
    assert len(enc.body) == 24 and [b.stride for blk in H.get_blocks(50) for b in blk].count(2) == 4
    with pytest.raises(ValueError):
        H.get_blocks(34)
    assert get_keys({"state_dict": {"encoder.a.b": 1, "decoder.c": 2}}, "encoder") == {"a.b": 1}
    opts = types.SimpleNamespace(output_size=256, encoder_type="GradualStyleEncoder", train_decoder=False, use_attention=True, pt_ckpt_path=None,
                                 stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True)
    net = pSp(opts)
    assert opts.n_styles == 14 and len(net.encoder.styles) == 14
    assert not any(p.requires_grad for p in net.decoder.parameters()) and all(p.requires_grad for p in net.encoder.parameters())
    with pytest.raises(Exception):
        pSp(types.SimpleNamespace(**{**vars(opts), "encoder_type": "nope"}))


def test_c_abi_argument_validation_without_a_gpu():
    """Error convention of the boundary: bad arguments come back as status codes before anything is launched --
    null pointers, non-positive sizes, unsupported modes -- so this runs on a machine without a GPU."""
    from face_mask_inpaint_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libfmi_hip.so not built")
    c = ctypes.CDLL(_lib.LIB_PATH)
    BAD, UNSUP = 1, 2
    i64, f32, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    c.fmi_gemm_f32.argtypes = [vp, vp, vp] + [ctypes.c_int] * 3 + [i64] * 6 + [ctypes.c_int] + [i64] * 3 + [f32, f32, vp, vp]
    assert c.fmi_gemm_f32(None, None, None, 4, 4, 4, 4, 1, 4, 1, 4, 1, 1, 0, 0, 0, 1.0, 0.0, None, None) == BAD
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, vp)
    assert c.fmi_gemm_f32(p, p, p, 0, 4, 4, 4, 1, 4, 1, 4, 1, 1, 0, 0, 0, 1.0, 0.0, None, None) == BAD          # M = 0
    assert c.fmi_gemm_f32(p, p, p, 4, 4, 4, 4, 2, 4, 1, 4, 1, 1, 0, 0, 0, 1.0, 0.0, None, None) == UNSUP        # A strided in both dims
    d = _lib.ConvDesc(N=1, H=8, W=8, C=4, OH=8, OW=8, K=4, x_cstride=4, y_cstride=4, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    c.fmi_conv2d_fwd_f32.argtypes = [ctypes.POINTER(_lib.ConvDesc), vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, i64, vp]
    assert c.fmi_conv2d_fwd_f32(ctypes.byref(d), None, p, None, None, p, 0, 1, 0, None) == BAD                 # x = NULL
    assert c.fmi_conv2d_fwd_f32(ctypes.byref(d), p, p, None, None, p, 7, 1, 0, None) == BAD                    # unknown activation
    d_bad = _lib.ConvDesc(N=1, H=8, W=8, C=4, OH=7, OW=8, K=4, x_cstride=4, y_cstride=4, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    assert c.fmi_conv2d_fwd_f32(ctypes.byref(d_bad), p, p, None, None, p, 0, 1, 0, None) != 0                  # OH inconsistent with H, pad, kh
    c.fmi_conv2d_thin_supported.argtypes = [ctypes.POINTER(_lib.ConvDesc)]
    assert c.fmi_conv2d_thin_supported(ctypes.byref(d)) == 1
    d5 = _lib.ConvDesc(N=1, H=8, W=8, C=4, OH=8, OW=8, K=5, x_cstride=4, y_cstride=5, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    assert c.fmi_conv2d_thin_supported(ctypes.byref(d5)) == 0
    c.fmi_upfirdn2d_f32.argtypes = [vp] * 3 + [ctypes.c_int] * 13 + [vp]
    assert c.fmi_upfirdn2d_f32(p, p, p, 1, 4, 4, 4, 4, 0, 1, 1, 1, 0, 0, 0, 0, None) == BAD                     # up_x = 0
    assert c.fmi_upfirdn2d_f32(p, p, p, 1, 2, 2, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == BAD                     # output would be empty
    c.fmi_avgpool_f32.argtypes = [vp, vp] + [ctypes.c_int] * 5 + [vp]
    assert c.fmi_avgpool_f32(p, p, 1, 4, 4, 4, 8, None) == BAD                                                 # window larger than the image
    # piece images: a weight whose shape the preparation kernel cannot cut into pieces must be refused, not left uninitialised
    we = _lib.WeightEntry(w=ctypes.addressof(buf), wf=ctypes.addressof(buf), wt=ctypes.addressof(buf), rows=16, C=16, taps=49, iters=0,
                          wf3=ctypes.addressof(buf))
    c.fmi_weight_prepare_f32.argtypes = [ctypes.POINTER(_lib.WeightEntry), ctypes.c_int, vp]
    assert c.fmi_weight_prepare_f32(ctypes.byref(we), 1, None) == UNSUP                                         # 7 x 7 taps: no 8-channel group fits a tile
    we2 = _lib.WeightEntry(w=ctypes.addressof(buf), wf=ctypes.addressof(buf), wt=ctypes.addressof(buf), rows=12, C=16, taps=9, iters=0,
                           wt3=ctypes.addressof(buf))
    assert c.fmi_weight_prepare_f32(ctypes.byref(we2), 1, None) == UNSUP                                        # rows % 8 != 0
    c.fmi_split3_f32.argtypes = [vp, vp, vp, i64, ctypes.c_int, ctypes.c_int, f32, vp]
    assert c.fmi_split3_f32(p, p, None, 4, 24, 0, 0.0, None) == BAD                                            # channels % 16 != 0
    assert c.fmi_split3_f32(p, p, None, 4, 16, 5, 0.0, None) == BAD                                            # unknown op
    d16 = _lib.ConvDesc(N=1, H=8, W=8, C=16, OH=8, OW=8, K=24, x_cstride=16, y_cstride=24, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    d16.y3 = ctypes.addressof(buf)
    assert c.fmi_conv2d_fwd_f32(ctypes.byref(d16), p, p, None, None, p, 0, 1, 0, None) == BAD                  # piece image of a 24-channel result
    c.fmi_resample_u8.argtypes = [vp, vp] + [ctypes.c_int] * 8 + [vp, vp, ctypes.c_int, vp]
    assert c.fmi_resample_u8(p, p, 1, 8, 8, 3, 4, 2, 0, 8, p, p, 5, None) == BAD                               # axis must be 0 or 1
    assert c.fmi_resample_u8(p, p, 1, 8, 8, 3, 4, 0, 4, 8, p, p, 5, None) == BAD                               # rows beyond the image
    c.fmi_ssim_valid_f32.argtypes = [vp, vp, vp] + [ctypes.c_int] * 4 + [f32, f32, vp, vp, i64, vp]
    assert c.fmi_ssim_valid_f32(p, p, p, 11, 1, 8, 8, 1e-4, 9e-4, p, p, 128, None) == BAD                      # image smaller than the window
    c.fmi_adam_step_dev_guarded_f32.argtypes = [vp, ctypes.c_int] + [f32] * 5 + [vp, vp, vp]
    assert c.fmi_adam_step_dev_guarded_f32(None, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, p, p, None) == BAD
    # fused StyledConv entries of the bf16 decoder (round 3)
    ci = ctypes.c_int
    c.fmi_blur_act_bf16.argtypes = [vp, vp, vp] + [ci] * 8 + [vp, vp, vp, vp, f32, f32, ci, vp]
    assert c.fmi_blur_act_bf16(None, p, p, 1, 9, 9, 64, 1, 1, 1, 1, None, None, None, None, 0.2, 1.4, 1, None) == BAD     # in = NULL
    assert c.fmi_blur_act_bf16(p, p, p, 1, 9, 9, 64, 1, 1, 1, 1, None, p, None, None, 0.2, 1.4, 1, None) == BAD        # noise without its weight
    assert c.fmi_blur_act_bf16(p, p, p, 1, 2, 2, 64, 0, 0, 0, 0, None, None, None, None, 1.0, 1.0, 1, None) == BAD     # 2 x 2 input: no output pixel
    assert c.fmi_blur_act_bf16(p, p, p, 1, 9, 9, 24, 1, 1, 1, 1, None, None, None, None, 1.0, 1.0, 1, None) == UNSUP   # channels % 32 != 0
    assert c.fmi_blur_act_bf16(p, p, p, 1, 9, 9, 64, 1, 1, 1, 1, None, None, None, p, 0.2, 1.4, 0, None) == UNSUP      # output stage needs rank-one taps
    c.fmi_styled_out_bwd_bf16.argtypes = [vp] * 11 + [i64, vp, ci, i64, ci, f32, f32, vp]
    assert c.fmi_styled_out_bwd_bf16(p, p, None, None, None, None, p, None, None, None, p, 16, p, 1, 4, 64, 0.2, 1.4, None) == BAD   # workspace < N*3*C
    assert c.fmi_styled_out_bwd_bf16(p, p, p, None, None, None, p, None, None, None, p, 4096, p, 1, 4, 64, 0.2, 1.4, None) == BAD   # noise without its weight
    assert c.fmi_styled_out_bwd_bf16(p, p, None, None, None, None, p, None, None, None, p, 4096, p, 1, 4, 12, 0.2, 1.4, None) == UNSUP  # channels % 8 != 0
    db = _lib.ConvDesc(N=1, H=8, W=8, C=64, OH=8, OW=8, K=128, x_cstride=64, y_cstride=128, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    c.fmi_conv2d_fwd_act_bf16.argtypes = [ctypes.POINTER(_lib.ConvDesc), vp, vp, vp, vp, vp, vp, f32, f32, vp, vp]
    assert c.fmi_conv2d_fwd_act_bf16(ctypes.byref(db), p, p, None, p, None, None, 0.2, 1.4, p, None) == BAD             # noise without its weight
    assert c.fmi_conv2d_fwd_act_bf16(ctypes.byref(db), None, p, None, None, None, None, 0.2, 1.4, p, None) == BAD       # x = NULL
    db32 = _lib.ConvDesc(N=1, H=8, W=8, C=32, OH=8, OW=8, K=128, x_cstride=32, y_cstride=128, kh=3, kw=3, stride=1, pad=1, pad_mode=0)
    assert c.fmi_conv2d_fwd_act_bf16(ctypes.byref(db32), p, p, None, None, None, None, 0.2, 1.4, p, None) == UNSUP      # 32 input channels: no eight-phase kernel
    c.fmi_debug_bf16_tile.argtypes = [ci]
    prev = c.fmi_debug_bf16_tile(-1)                                                                                 # query only
    assert c.fmi_debug_bf16_tile(8) == prev and c.fmi_debug_bf16_tile(prev) == 8 and c.fmi_debug_bf16_tile(-1) == prev


def test_pmc_traffic_profile_is_current():
    """bench.py's roofline.traffic comes from profiles/pmc_traffic.json; every entry names the kernel it was collected on, the raw
    PMC summary it was taken from and the sha256 of the kernel's source at that time.  A kernel edited after its PMC pass makes the
    number stale: this test fails (and bench.py reports traffic = null with the reason) until the pass is repeated."""
    import hashlib
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tab = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert tab, "no PMC traffic entries"
    for key, ent in tab.items():
        src = os.path.join(root, ent["source"])
        assert hashlib.sha256(open(src, "rb").read()).hexdigest() == ent["source_sha256"], \
            f"{key}: {ent['source']} changed since {ent['profile']} was collected -- repeat the rocprofv3 --pmc passes (tools/bench_tools/pmc_collect.sh)"
        raw = json.load(open(os.path.join(root, ent["profile"])))
        names = {e["kernel"] for sect in ("fetch", "write") for e in raw[sect]}
        assert ent["kernel"] in names, f"{key}: kernel {ent['kernel']!r} is not in {ent['profile']}"
        fetch = [e["mean_value"] for e in raw["fetch"] if e["kernel"] == ent["kernel"]][0]
        write = [e["mean_value"] for e in raw["write"] if e["kernel"] == ent["kernel"]][0]
        assert abs(fetch - ent["fetch_kb"]) < 1e-3 * fetch and abs(write - ent["write_kb"]) < 1e-3 * max(write, 1.0)
