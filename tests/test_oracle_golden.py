"""CPU: the oracle (oracle/picnet_cpu.py) against the golden vectors produced by the imported
reference (oracle/gen_golden.py).  This is what pins the oracle."""
import torch
import pytest

from oracle import picnet_cpu as O

TOL = dict(rtol=1e-5, atol=1e-6)


def _params(sd):
    return O.prepare_params(sd)


def _check_block(fx, fn):
    P = _params(fx["sd0"])
    xs = [x.clone().requires_grad_(True) for x in fx["inputs"]]
    y = fn(P, *xs)
    torch.testing.assert_close(y, fx["out"], **TOL)
    y.backward(fx["gout"])
    for x, g in zip(xs, fx["gin"]):
        if g.numel():
            torch.testing.assert_close(x.grad, g, rtol=1e-4, atol=1e-6)
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P[n].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")
    for k, v in fx["sd1"].items():  # SpectralNorm u/v state after one forward
        if k.endswith("weight_u") or k.endswith("weight_v"):
            torch.testing.assert_close(P[k], v, **TOL)


def test_resblock_none(golden):
    _check_block(golden("picnet_ops.pt")["resblock_none"], lambda P, x: O.res_block(_strip(P), "b", x, "none"))


def _strip(P):
    return {"b." + k: v for k, v in P.items()}


@pytest.mark.parametrize("name,fn", [
    ("resblock_down", lambda P, x: O.res_block(P, "b", x, "down")),
    ("resblock_enc_opt", lambda P, x: O.res_block_encoder_optimized(P, "b", x)),
    ("resblock_dec", lambda P, x: O.res_block_decoder(P, "b", x)),
    ("output", lambda P, x: O.output_block(P, "b", x)),
    ("auto_attn", lambda P, x: O.auto_attn(P, "b", x)),
    ("ex_guided_att", lambda P, m, s, r: O.example_guided_attention(P, "b", m, s, r)),
    ("ex_guided_att_out", lambda P, m, s, r: O.example_guided_attention(P, "b", m, s, r)),
])
def test_blocks(golden, name, fn):
    fx = dict(golden("picnet_ops.pt")[name])
    fx["sd0"] = {"b." + k: v for k, v in fx["sd0"].items()}
    fx["sd1"] = {"b." + k: v for k, v in fx["sd1"].items()}
    fx["gparams"] = {"b." + k: v for k, v in fx["gparams"].items()}
    _check_block(fx, fn)


def test_functional_pieces(golden):
    fx = golden("picnet_ops.pt")
    torch.testing.assert_close(O.gram_matrix(fx["gram"]["x"]), fx["gram"]["out"], **TOL)
    for name, fn in (("style_loss", O.style_loss), ("contextual_loss", O.contextual_loss)):
        x = fx[name]["x"].clone().requires_grad_(True)
        l = fn(x, fx[name]["y"])
        torch.testing.assert_close(l, fx[name]["out"], **TOL)
        l.backward()
        torch.testing.assert_close(x.grad, fx[name]["gx"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(O.lsgan(fx["lsgan"]["pred"], True), fx["lsgan"]["real"], **TOL)
    torch.testing.assert_close(O.lsgan(fx["lsgan"]["pred"], False), fx["lsgan"]["fake"], **TOL)
    torch.testing.assert_close(O.scale_img(fx["scale_img"]["mask"], (4, 4)), fx["scale_img"]["out"], **TOL)
    torch.testing.assert_close(O.scale_img(fx["scale_img"]["mask"], (5, 7)), fx["scale_img"]["out_odd"], **TOL)
    assert torch.equal(O.binarise_mask(fx["binarise"]["mask"]), fx["binarise"]["out"])  # bit-exact


def test_two_training_steps(golden):
    """Whole path: ReferenceFill forward + GANOptimizer.__call__ for two consecutive steps
    (SpectralNorm u/v evolution, three D calls per step, Adam updates)."""
    fx = golden("picnet_train_tiny.pt")
    cfg = fx["config"]
    PG, PD = O.prepare_params(fx["G_sd0"]), O.prepare_params(fx["D_sd0"])
    PV = O.prepare_params(fx["V_sd"], frozen=True)
    opt_g = torch.optim.Adam(O.unique_trainable(PG), lr=cfg["lr"])
    opt_d = torch.optim.Adam(O.unique_trainable(PD), lr=cfg["lr"])
    kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"],
              dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)

    for step in range(2):
        s = fx[f"step{step}"]
        mask = O.binarise_mask(s["mask"])
        gen = O.reference_fill_forward(PG, s["src"], s["ref"], mask, s["eps_p"], s["eps_q"], **kw)
        torch.testing.assert_close(gen, s["gen"], rtol=1e-4, atol=1e-5)
        PDl = {("" if not k else k): v for k, v in PD.items()}
        g_loss, perc, sty, cx = _gen_losses(PDl, PV, s, gen, mask, cfg)
        opt_g.zero_grad()
        g_loss.backward()
        for n, g in s["G_grads"].items():
            torch.testing.assert_close(PG[n].grad, g, rtol=2e-3, atol=1e-7, msg=lambda m, n=n: f"G grad {n}: {m}")
        opt_g.step()
        d_loss = (O.lsgan(O.res_discriminator(PD, "", s["gt"], cfg["disc_layers"]), True)
                  + O.lsgan(O.res_discriminator(PD, "", gen.detach(), cfg["disc_layers"]), False)) * 0.5
        opt_d.zero_grad()
        d_loss.backward()
        for n, g in s["D_grads"].items():
            torch.testing.assert_close(PD[n].grad, g, rtol=2e-3, atol=1e-7, msg=lambda m, n=n: f"D grad {n}: {m}")
        opt_d.step()
        for got, key in ((g_loss, "g_loss"), (d_loss, "d_loss"), (perc, "perc"), (sty, "style"), (cx, "cx")):
            torch.testing.assert_close(got.detach(), s[key], rtol=1e-4, atol=1e-9)
    for k, v in fx["G_sd2"].items():
        torch.testing.assert_close(PG[k].detach(), v, rtol=1e-4, atol=2e-6, msg=lambda m, k=k: f"G param {k}: {m}")
    for k, v in fx["D_sd2"].items():
        torch.testing.assert_close(PD[k].detach(), v, rtol=1e-4, atol=2e-6, msg=lambda m, k=k: f"D param {k}: {m}")


def test_two_training_steps_float64(golden):
    """the same restatement evaluated in float64, each step restarted from the reference's fp32 state at the start of that step,
    against the reference's own float64 evaluation (gen_golden.py: G_grads64 / D_grads64, stored rounded to fp32): float64 has no
    kink flips to speak of, so the two agree to fp32 storage precision -- this pins the adjudicator of the GPU gradient tests"""
    fx = golden("picnet_train_tiny.pt")
    cfg = fx["config"]
    kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"],
              dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)
    dt = torch.float64
    for step in range(2):
        s = {k: (v.to(dt) if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in fx[f"step{step}"].items()}
        PG = O.prepare_params(fx[f"G_sd{step}"], dtype=dt)
        PD = O.prepare_params(fx[f"D_sd{step}"], dtype=dt)
        PV = O.prepare_params(fx["V_sd"], frozen=True, dtype=dt)
        mask = O.binarise_mask(s["mask"]).to(dt)
        gen = O.reference_fill_forward(PG, s["src"], s["ref"], mask, s["eps_p"], s["eps_q"], **kw)
        torch.testing.assert_close(gen.float(), fx[f"step{step}"]["gen64"], rtol=1e-6, atol=1e-7)
        g_loss, perc, sty, cx = _gen_losses(PD, PV, s, gen, mask, cfg)
        g_loss.backward()
        for n, g in s["G_grads64"].items():
            lim = 1e-6 * float(g.abs().max()) + 1e-12
            assert float((PG[n].grad - g.to(dt)).abs().max()) <= lim, f"G grad {n} step {step}"
        for t in PD.values():
            t.grad = None
        d_loss = (O.lsgan(O.res_discriminator(PD, "", s["gt"], cfg["disc_layers"]), True)
                  + O.lsgan(O.res_discriminator(PD, "", gen.detach(), cfg["disc_layers"]), False)) * 0.5
        d_loss.backward()
        for n, g in s["D_grads64"].items():
            lim = 1e-6 * float(g.abs().max()) + 1e-12
            assert float((PD[n].grad - g.to(dt)).abs().max()) <= lim, f"D grad {n} step {step}"
        got = torch.stack([d_loss, g_loss, perc, sty, cx]).detach()
        torch.testing.assert_close(got, s["losses64"], rtol=1e-6, atol=1e-15)  # an fp32-rounded constant (1e-5, 0.5) on one side: 3e-8


def _gen_losses(PD, PV, s, gen, mask, cfg):
    import torch.nn.functional as F

    g = O.lsgan(O.res_discriminator(PD, "", gen, cfg["disc_layers"]), True) * O.LAMBDA_G + F.l1_loss(gen, s["gt"])
    perc = O.vgg_loss(PV, "", gen, s["gt"], "perceptual") * O.LAMBDA_PERC
    sty = O.vgg_loss(PV, "", gen * (1 - mask).unsqueeze(1), s["src"], "style") * O.LAMBDA_STYLE
    cx = O.vgg_loss(PV, "", gen * mask.unsqueeze(1), s["ref"] * mask.unsqueeze(1), "contextual") * O.LAMBDA_CX
    return g + perc + sty + cx, perc, sty, cx


# ---- StyleGAN2 decoder pieces ------------------------------------------------------------------------------
def test_stylegan2_native_ops_oracle(golden):
    """C restatement of upfirdn2d / fused_bias_act against the reference's own upfirdn2d_native outputs"""
    from oracle import stylegan2_cpu as S

    fx = golden("stylegan2_ops.pt")
    for case in fx["upfirdn2d"]:
        px0, px1, py0, py1 = case["pad"]
        got = S.upfirdn2d_planes(case["x"], case["k"], case["up"], case["up"], case["down"], case["down"], px0, px1, py0, py1)
        torch.testing.assert_close(got, case["out"], rtol=1e-5, atol=1e-6)
        if px0 == py0 and px1 == py1:
            got_t = S.upfirdn2d_t(case["x"].unsqueeze(0), case["k"], case["up"], case["down"], (px0, px1))[0]
            torch.testing.assert_close(got_t, case["out"], rtol=1e-5, atol=1e-6)
    f = fx["fused_lrelu"]
    torch.testing.assert_close(S.fused_leaky_relu(f["x"], f["b"]), f["out"], rtol=1e-6, atol=1e-7)
    m = torch.randint(-5, 300, (3, 7, 9))
    assert torch.equal(S.mask_binarise(m), (m > 0).float())


def test_stylegan2_blocks_oracle(golden):
    from oracle import stylegan2_cpu as S

    fx = golden("stylegan2_ops.pt")
    for name, up, demod in (("modconv", False, True), ("modconv_up", True, True), ("modconv_rgb", False, False)):
        f = fx[name]
        P = {"m." + k: v.clone().requires_grad_(True) for k, v in f["sd"].items() if v.dtype.is_floating_point}
        x, s = f["x"].clone().requires_grad_(True), f["style"].clone().requires_grad_(True)
        y = S.modulated_conv(P, "m", x, s, demodulate=demod, upsample=up)
        torch.testing.assert_close(y, f["out"], rtol=1e-4, atol=1e-5)
        y.backward(f["gout"])
        torch.testing.assert_close(x.grad, f["gx"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(s.grad, f["gstyle"], rtol=1e-4, atol=1e-5)
        for n, g in f["gparams"].items():
            torch.testing.assert_close(P["m." + n].grad, g, rtol=1e-4, atol=1e-5)
    f = fx["styledconv_up"]
    P = {"m." + k: v.clone() for k, v in f["sd"].items() if v.dtype.is_floating_point}
    torch.testing.assert_close(S.styled_conv(P, "m", f["x"], f["style"], f["noise"], upsample=True), f["out"], rtol=1e-4, atol=1e-5)
    f = fx["torgb"]
    P = {"m." + k: v.clone() for k, v in f["sd"].items() if v.dtype.is_floating_point}
    torch.testing.assert_close(S.to_rgb(P, "m", f["x"], f["style"], f["skip"]), f["out"], rtol=1e-4, atol=1e-5)


def test_ssim_oracle(golden):
    from oracle import ssim_cpu as S

    fx = golden("ssim.pt")
    torch.testing.assert_close(S.ssim(fx["a"], fx["b"]), fx["mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(S.ssim(fx["a"], fx["b"], size_average=False), fx["per_image"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(S.ssim(fx["a"], fx["a"]), fx["same"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# pSp encoder side (rows B1, B9): oracle/psp_cpu.py vs the reference's own modules
# ---------------------------------------------------------------------------------------------------------------------
def _psp_params(sd, prefix="b."):
    P = {}
    for k, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
        P[prefix + k] = v
    return P


PSP_BLOCKS = [("ir_se_conv_s2", 2), ("ir_se_pool_s1", 1), ("ir_se_pool_s2", 2), ("ir_conv_s2", 2)]


@pytest.mark.parametrize("name,stride", PSP_BLOCKS)
def test_psp_bottlenecks_oracle(golden, name, stride):
    from oracle import psp_cpu as PO
    fx = golden("psp_ops.pt")[name]
    P = _psp_params(fx["sd"])
    x = fx["x"].clone().requires_grad_(True)
    y = PO.bottleneck(P, "b", x, stride, training=True)
    torch.testing.assert_close(y, fx["out"], rtol=1e-5, atol=2e-6)
    y.backward(fx["gout"])
    torch.testing.assert_close(x.grad, fx["gx"], rtol=1e-4, atol=2e-6)
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P["b." + n].grad, g, rtol=1e-4, atol=1e-4 * float(g.abs().max()) + 1e-7, msg=lambda m, n=n: f"{n}: {m}")
    for k, v in fx["stats_after"].items():
        torch.testing.assert_close(P["b." + k], v, rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    P = _psp_params({**fx["sd"], **fx["stats_after"]})  # the reference's eval pass ran after the training pass
    with torch.no_grad():
        torch.testing.assert_close(PO.bottleneck(P, "b", fx["x"], stride, training=False), fx["out_eval"], rtol=1e-5, atol=2e-6)


def test_psp_style_block_oracle(golden):
    from oracle import psp_cpu as PO
    fx = golden("psp_ops.pt")["style_block"]
    P = _psp_params(fx["sd"])
    x = fx["x"].clone().requires_grad_(True)
    y = PO.gradual_style_block(P, "b", x, 16)
    torch.testing.assert_close(y, fx["out"], **TOL)
    y.backward(fx["gout"])
    torch.testing.assert_close(x.grad, fx["gx"], rtol=1e-4, atol=1e-6)
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P["b." + n].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")


def test_psp_encoder_oracle(golden):
    from oracle import psp_cpu as PO
    fx = golden("psp_ops.pt")["encoder"]
    P = _psp_params(fx["sd"], "")
    x, ref = fx["x"].clone().requires_grad_(True), fx["ref"].clone().requires_grad_(True)
    out = PO.gradual_style_encoder(P, "", x, ref, fx["mask"], fx["n_styles"], True, True)
    torch.testing.assert_close(out, fx["out"], rtol=1e-4, atol=1e-5)
    out.backward(fx["gout"])
    torch.testing.assert_close(x.grad, fx["gx"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(ref.grad, fx["gref"], rtol=1e-3, atol=1e-5)
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P[n].grad, g, rtol=1e-3, atol=1e-3 * float(g.abs().max()) + 1e-7, msg=lambda m, n=n: f"{n}: {m}")
    for k, v in fx["stats_after"].items():
        torch.testing.assert_close(P[k], v, rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    P = _psp_params({**fx["sd"], **fx["stats_after"]}, "")
    with torch.no_grad():
        torch.testing.assert_close(PO.gradual_style_encoder(P, "", fx["x"], fx["ref"], fx["mask"], fx["n_styles"], True, False), fx["out_eval"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(PO.gradual_style_encoder(P, "", fx["x"], None, None, fx["n_styles"], True, False), fx["out_eval_noref"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(PO.gradual_style_encoder(P, "", fx["x"], fx["ref"], fx["mask"], fx["n_styles"], False, False), fx["out_eval_noatt"], rtol=1e-4, atol=1e-5)


def test_psp_loss_oracle(golden):
    from oracle import psp_cpu as PO
    fx = golden("psp_ops.pt")["psp_loss"]
    a = fx["args"]
    yh, lat = fx["y_hat"].clone().requires_grad_(True), fx["latent"].clone().requires_grad_(True)
    loss = PO.psp_loss(fx["y"], yh, lat, fx["latent_avg"], fx["ref"], fx["mask"], a["l2_lambda"], a["l2_lambda_ref"], a["w_norm_lambda"])
    torch.testing.assert_close(loss, fx["loss"], **TOL)
    loss.backward()
    torch.testing.assert_close(yh.grad, fx["gy_hat"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(lat.grad, fx["glatent"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(PO.psp_loss(fx["y"], fx["y_hat"], fx["latent"], None, None, None, a["l2_lambda"], a["l2_lambda_ref"], a["w_norm_lambda"]),
                               fx["loss_nomask"], **TOL)
    # the logged-only VGG terms (criteria/__init__.py:74-76,88-90)
    PV = O.prepare_params({"vgg" + k: v for k, v in fx["vgg"].items()}, frozen=True)
    m = fx["mask"].unsqueeze(1)
    with torch.no_grad():
        st = O.vgg_loss(PV, "vgg", fx["y_hat"] * (1 - m), fx["x"], "style") * a["style_lambda"]
        cx = O.vgg_loss(PV, "vgg", fx["y_hat"] * m, fx["ref"] * m, "contextual") * a["cx_lambda"]
    torch.testing.assert_close(st, fx["loss_dict"]["loss_style"].float(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(cx, fx["loss_dict"]["loss_context"].float(), rtol=1e-4, atol=1e-7)


def test_patch_discriminator_oracle(golden):
    """--disc_model_type PatchDis (network.py:373-430), the alternative of row A8"""
    fx = golden("picnet_patchdis.pt")
    P = _params(fx["sd0"])
    x = fx["x"].clone().requires_grad_(True)
    y = O.patch_discriminator(P, "", x)
    torch.testing.assert_close(y, fx["out"], **TOL)
    y.backward(fx["gout"])
    torch.testing.assert_close(x.grad, fx["gx"], rtol=1e-4, atol=1e-6)
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P[n].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")
    for k, v in fx["sd1"].items():
        if k.endswith("weight_u") or k.endswith("weight_v"):
            torch.testing.assert_close(P[k], v, **TOL)


# ---- whole models: parameters come from oracle/seeded.py on both sides, the fixture holds I/O and gradient digests -------------
def _leaf_params(module, trainable=True):
    """flat name -> leaf tensor dictionary (the oracle's interface) of a module filled by seeded_fill_"""
    P = {}
    for k, v in module.state_dict().items():
        t = v.detach().clone()
        if trainable and t.is_floating_point() and not (k.endswith(".kernel") or "running_" in k or k.startswith("noises.") or ".noises." in k):
            t.requires_grad_(True)
        P[k] = t
    return P


def test_generator_whole_oracle(golden):
    """oracle generator_forward / generator_styles_forward against the WHOLE reference Generator(64, 512, 2)
    (tests/golden/stylegan2_generator.pt, oracle/gen_golden.py:generator_fixture): latent indexing, noise order, skip accumulation,
    mapping network, style mixing, truncation -- with parameter gradients"""
    from face_mask_inpaint_amd.modules.psp.stylegan2.model import Generator  # parameter container only (no forward on the CPU)
    from oracle import stylegan2_cpu as S
    from oracle.seeded import as_digest, check_adjudicated, check_digest, seeded_fill_, seeded_tensor

    fx = golden("stylegan2_generator.pt")
    cfg = fx["config"]
    gen = Generator(cfg["size"], cfg["style_dim"], cfg["n_mlp"])
    seeded_fill_(gen, cfg["seed"])
    P = _leaf_params(gen)
    noises = [P[f"noises.noise_{i}"] for i in range(gen.num_layers)]
    c = fx["wplus"]
    lat = seeded_tensor((2, gen.n_latent, 512), c["latent_seed"]).requires_grad_(True)
    img = S.generator_forward(P, lat, noises, cfg["size"])
    torch.testing.assert_close(img, c["image"], rtol=1e-4, atol=1e-4)
    (img * seeded_tensor(img.shape, c["cot_seed"])).sum().backward()
    g32, g64 = dict(c["gparams"], glatent=as_digest(c["glatent"])), dict(c["gparams64"], glatent=as_digest(c["glatent64"]))
    check_adjudicated(dict({n: P[n].grad for n in c["gparams64"]}, glatent=lat.grad), g32, g64, what="Generator W+")
    assert sorted(n for n, _ in gen.named_parameters() if P[n].grad is None) == c["no_grad"]  # the mapping network is unused here
    for t in P.values():
        t.grad = None
    m = fx["mix"]
    z1 = seeded_tensor((2, 512), m["z_seeds"][0]).requires_grad_(True)
    z2 = seeded_tensor((2, 512), m["z_seeds"][1])
    tl = seeded_tensor((1, 512), m["trunc_seed"], 0.5)
    nz = [seeded_tensor(P[f"noises.noise_{i}"].shape, m["noise_seed0"] + i) for i in range(gen.num_layers)]
    img, latent, feat = S.generator_styles_forward(P, [z1, z2], nz, cfg["size"], cfg["n_mlp"], inject_index=m["inject_index"],
                                                   truncation=m["truncation"], truncation_latent=tl)
    torch.testing.assert_close(img, m["image"], rtol=1e-4, atol=1e-4)
    check_digest(feat, m["feature"], 1e-5, "feature")
    (img * seeded_tensor(img.shape, m["cot_seed"])).sum().backward()
    g32, g64 = dict(m["gparams"], gz1=as_digest(m["gz1"])), dict(m["gparams64"], gz1=as_digest(m["gz164"]))
    check_adjudicated(dict({n: P[n].grad for n in m["gparams64"]}, gz1=z1.grad), g32, g64, what="Generator mapping network")
    with torch.no_grad():
        torch.testing.assert_close(S.mapping_network(P, seeded_tensor((4, 512), fx["mean_latent_input"]["seed"]), cfg["n_mlp"]),
                                   fx["mean_latent_input"]["out"], rtol=1e-5, atol=1e-5)


def _psp_whole(cfg):
    import types

    from face_mask_inpaint_amd.modules.psp.psp import pSp  # parameter container
    from oracle.seeded import seeded_fill_, seeded_tensor

    opts = types.SimpleNamespace(output_size=cfg["output_size"], encoder_type="GradualStyleEncoder", use_attention=True, train_decoder=True,
                                 start_from_latent_avg=True, learn_in_w=False, pt_ckpt_path=None, stylegan_weights=None)
    net = pSp(opts)
    seeded_fill_(net, cfg["seed"])
    net.latent_avg = seeded_tensor((opts.n_styles, 512), cfg["latent_avg_seed"], 0.5)
    x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["x_seed"])) * 2 - 1
    ref = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["ref_seed"])) * 2 - 1
    mask = torch.zeros(2, 256, 256)
    for i, (a, b, c, d) in enumerate(cfg["rects"]):
        mask[i, a:b, c:d] = 1
    return net, x, ref, mask


def test_psp_whole_oracle(golden):
    """oracle psp_forward against the WHOLE reference pSp in training mode at full widths (tests/golden/psp_whole.pt,
    gen_golden.py:psp_whole_fixture): forward + backward incl. train-mode BatchNorm, latent_avg, noise buffers"""
    from oracle import psp_cpu as PS
    from oracle.seeded import check_adjudicated, check_digest, seeded_tensor

    fx = golden("psp_whole.pt")
    cfg = fx["config"]
    net, x, ref, mask = _psp_whole(cfg)
    P = _leaf_params(net)
    x.requires_grad_(True)
    ref.requires_grad_(True)
    nl = int(__import__("math").log2(cfg["output_size"]) - 2) * 2 + 1
    noises = [P[f"decoder.noises.noise_{i}"] for i in range(nl)]
    img, lat = PS.psp_forward(P, x, ref, mask, noises, cfg["output_size"], latent_avg=net.latent_avg, training=True)
    scale = float(fx["image"].abs().max())
    torch.testing.assert_close(lat, fx["latent"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(img, fx["image"], rtol=1e-3, atol=1e-4 * scale)
    ((img * seeded_tensor(img.shape, cfg["cot_seeds"][0])).sum() / 256.0 + (lat * seeded_tensor(lat.shape, cfg["cot_seeds"][1])).sum()).backward()
    check_adjudicated({"gx": x.grad, "gref": ref.grad}, {"gx": fx["gx"], "gref": fx["gref"]}, {"gx": fx["gx64"], "gref": fx["gref64"]},
                      floor=5e-3, what="pSp input gradients")
    check_adjudicated({n: P[n].grad for n in fx["gparams64"]}, fx["gparams"], fx["gparams64"], what="pSp parameters")
    assert sorted(n for n, _ in net.named_parameters() if P[n].grad is None) == fx["no_grad"]
    for k, v in fx["stats_after"].items():
        torch.testing.assert_close(P[k], v, rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        img_e, _ = PS.psp_forward(P, x.detach(), ref.detach(), mask, noises, cfg["output_size"], latent_avg=net.latent_avg, training=False)
    check_digest(img_e, fx["image_eval"], 1e-4, "eval image")
