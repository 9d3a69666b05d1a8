"""GPU: the drop-in nn.Modules (face_mask_inpaint_amd.modules) against the golden vectors produced by the imported
reference (tests/golden, oracle/gen_golden.py) and, at a larger size, against the CPU oracle.
Tolerance: 1e-3 relative on fp32 activations end to end (north_star); tighter per block."""
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


def _load(mod, sd, dev):
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # only the aliases of shared convs (model.N.module.* / shortcut.*) may be absent from the de-aliased fixture
    assert all(".shortcut." in "." + k or ".module." in k for k in missing), missing
    return mod.to(dev)


def _run_block(fx, mod, dev, rtol=1e-4, atol=1e-5, gtol=2e-4, const_inputs=()):
    mod = _load(mod, fx["sd0"], dev)
    xs = [x.to(dev).requires_grad_(i not in const_inputs) for i, x in enumerate(fx["inputs"])]
    y = mod(*xs)
    if isinstance(y, tuple):
        y = y[0]
    torch.testing.assert_close(y.detach().cpu(), fx["out"], rtol=rtol, atol=atol)
    y.backward(fx["gout"].to(dev))
    for i, (x, g) in enumerate(zip(xs, fx["gin"])):
        if g.numel() and i not in const_inputs:
            torch.testing.assert_close(x.grad.cpu(), g, rtol=gtol, atol=1e-5)
    params = dict(mod.named_parameters())
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(params[n].grad.cpu(), g, rtol=gtol, atol=2e-5, msg=lambda m, n=n: f"{n}: {m}")
    sd1 = mod.state_dict()
    for k, v in fx["sd1"].items():  # SpectralNorm u/v after one forward
        if k.endswith("weight_u") or k.endswith("weight_v"):
            torch.testing.assert_close(sd1[k].cpu(), v, rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")


def test_blocks_against_reference_golden(dev, golden):
    from torch import nn

    from face_mask_inpaint_amd.modules.example_guided_att import ExampleGuidedAttention
    from face_mask_inpaint_amd.modules.pluralistic_model import base_function as bf

    fx = golden("picnet_ops.pt")
    act = bf.get_nonlinearity_layer("LeakyReLU")
    inorm = bf.get_norm_layer("instance")
    _run_block(fx["resblock_none"], bf.ResBlock(8, 16, 8, None, act, "none", True, False), dev)
    _run_block(fx["resblock_down"], bf.ResBlock(8, 16, 8, None, act, "down", True, False), dev)
    _run_block(fx["resblock_enc_opt"], bf.ResBlockEncoderOptimized(3, 8, None, act, True, False), dev)
    _run_block(fx["resblock_dec"], bf.ResBlockDecoder(8, 4, 4, inorm, act, True, False), dev, gtol=1e-3)
    _run_block(fx["output"], bf.Output(8, 3, 3, None, act, True, False), dev)
    _run_block(fx["auto_attn"], bf.Auto_Attn(16, None), dev, gtol=5e-4)  # softmax backward: 2.4e-5 on one 0.06 entry of 960 (largest 2.8)
    _run_block(fx["ex_guided_att"], ExampleGuidedAttention(16), dev, const_inputs=(0,))  # the mask never carries a gradient
    _run_block(fx["ex_guided_att_out"], ExampleGuidedAttention(16, 16), dev, const_inputs=(0,))


def _tiny_models(fx, dev):
    from face_mask_inpaint_amd.modules.loss import GANOptimizer
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from face_mask_inpaint_amd.modules.pluralistic_model import network
    from face_mask_inpaint_amd.optim import FusedAdam

    cfg = fx["config"]
    enc = dict(type="pluralistic", ngf=8, z_nc=cfg["enc_z_nc"], img_f=16, layers=5, norm="none", activation="LeakyReLU", L=cfg["enc_L"])
    dec = dict(ngf=8, z_nc=16, img_f=32, layers=5, norm="instance", activation="LeakyReLU", L=0)
    G = _load(ReferenceFill(None, dict(enc), dict(dec), use_att=True, out_size=(cfg["out_size"],) * 2), fx["G_sd0"], dev)
    D = _load(network.define_d(ndf=8, img_f=32, layers=cfg["disc_layers"], norm="none", activation="LeakyReLU", model_type="ResDis"), fx["D_sd0"], dev)
    optG = FusedAdam([p for p in G.parameters() if p.requires_grad], lr=cfg["lr"])
    optD = FusedAdam([p for p in D.parameters() if p.requires_grad], lr=cfg["lr"])
    gopt = GANOptimizer(optD, optG, vgg_width_div=cfg["vgg_div"])
    gopt.vgg_loss.load_state_dict(fx["V_sd"])
    return G, D, gopt.to(dev), optG, optD


def _spy(opt, names_params, sink):
    orig = opt.step

    def step(closure=None):
        sink.append({n: p.grad.detach().cpu().clone() for n, p in names_params if p.grad is not None})
        return orig(closure)

    opt.step = step


def _reference_fp32_error(fx, key):
    """worst error of the REFERENCE's own fp32 gradients against its float64 evaluation, relative to each tensor's largest entry,
    over both golden steps (tensors that are analytically zero -- a conv bias in front of InstanceNorm -- excluded)"""
    worst = 0.0
    for step in (0, 1):
        st = fx[f"step{step}"]
        for n, g64 in st[f"{key}_grads64"].items():
            mx = float(g64.abs().max())
            if mx > 1e-12:
                worst = max(worst, float((st[f"{key}_grads"][n] - g64).abs().max()) / mx)
    return worst


def _check_grads_fp64(got, st, key, step, net_bound):
    """HIP gradients against the reference evaluated in float64 (the adjudicator, oracle/gen_golden.py).

    End-to-end gradients of this tiny random-init network differ between ANY two fp32 evaluations by LeakyReLU / ReLU kink flips:
    a pre-activation whose float64 value is below the forward rounding error (|x| ~ 1e-7, one or two among the millions of a step)
    lands on the other side of zero and changes one derivative from 1 to 0.1 (tools/bench_tools/kinks.py counts them: step 0 -- HIP
    flips two inputs of the Output block's LeakyReLU with |x64| = 1.2e-7 / 6.2e-8, the reference's fp32 run flips one of decoder4's;
    step 1 -- both flip the same element).  Which evaluation flips which element is chance, so the reference's fp32 run is 1e-5 from
    float64 at step 0 and 1.5e-3 at step 1, the HIP path 1.1e-3 and 1.5e-3.  Hence two bounds:
      * every tensor: error / max|g64| <= 2 x the WORST such ratio the reference's own fp32 run shows on this fixture (any tensor,
        either step) -- ~3e-3 for G; the round-1 bound was 4e-2;
      * per tensor against the reference's error on that same tensor (2 x + a 4e-6 floor): asserted for D (no VGG / decoder kinks
        in its short backward), reported for G."""
    worse, n_t = 0, 0
    for n, g64 in st[f"{key}_grads64"].items():
        assert n in got, f"missing grad {key}.{n}"
        mx = float(g64.abs().max())
        if mx <= 1e-12:  # analytically zero: rounding noise on both sides
            assert float(got[n].abs().max()) <= 1e-6, f"{key} grad {n} step {step} should vanish"
            continue
        err = float((got[n] - g64).abs().max()) / mx
        ref = float((st[f"{key}_grads"][n] - g64).abs().max()) / mx
        assert err <= net_bound, f"{key} grad {n} step {step}: {err:.3e} of max|g| > {net_bound:.3e} (reference fp32: {ref:.3e})"
        n_t += 1
        if err > 2 * ref + 4e-6:
            worse += 1
            assert key != "D", f"D grad {n} step {step}: {err:.3e} > 2 x reference's own {ref:.3e}"
    print(f"step {step} {key}: {worse} of {n_t} tensors further than 2 x the reference's own fp32 error from float64")


def test_two_training_steps_against_reference_golden(dev, golden):
    """ReferenceFill.forward + GANOptimizer.__call__ for two consecutive steps.

    Step 0 from the golden initial state: generated image and all five losses against the imported reference (1e-3), every
    parameter gradient (captured just before each optimiser step) against the reference's FLOAT64 evaluation with bounds derived
    from the reference's own fp32-vs-float64 error (see _check_grads_fp64).
    Step 1 two ways: (a) continued on this run's own post-step-0 state: image and losses against the golden (Adam's first update
    is lr * g / (|g| + 1e-8), so rounding noise in ~1e-8 gradients moves parameters by up to lr -- hence loose end-state bounds);
    (b) restarted from the REFERENCE's state at the start of step 1 (G_sd1 / D_sd1 incl. SpectralNorm u/v): gradients against
    float64 as in step 0."""
    from face_mask_inpaint_amd import functional as FF

    fx = golden("picnet_train_tiny.pt")
    cfg = fx["config"]
    bound_g = 2 * _reference_fp32_error(fx, "G")
    bound_d = max(2 * _reference_fp32_error(fx, "D"), 1e-5)
    assert bound_g < 5e-3 and bound_d < 1e-4, (bound_g, bound_d)

    def run_step(G, D, gopt, s):
        m = FF.binarise_mask(s["mask"].to(dev))
        assert torch.equal(m.cpu(), (s["mask"] > 0).float())
        gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
        return gen, gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)

    def check_outputs(gen, losses, s, step):
        torch.testing.assert_close(gen.detach().cpu(), s["gen"], rtol=1e-3, atol=1e-5)
        d_loss, g_loss, perc, sty, cx = losses
        for got, key in ((g_loss, "g_loss"), (d_loss, "d_loss"), (perc, "perc"), (sty, "style"), (cx, "cx")):
            torch.testing.assert_close(got.detach().cpu(), s[key], rtol=1e-3, atol=1e-9, msg=lambda mm, key=key: f"{key} step {step}: {mm}")
        for got, want in zip(losses, s["losses64"]):  # and the float64 values
            assert abs(float(got) / float(want) - 1) <= 1e-3

    # ---- step 0, then step 1 on this run's own state
    G, D, gopt, optG, optD = _tiny_models(fx, dev)
    grads = {"G": [], "D": []}
    _spy(optG, list(G.named_parameters()), grads["G"])
    _spy(optD, list(D.named_parameters()), grads["D"])
    gen, losses = run_step(G, D, gopt, fx["step0"])
    check_outputs(gen, losses, fx["step0"], 0)
    _check_grads_fp64(grads["G"][0], fx["step0"], "G", 0, bound_g)
    _check_grads_fp64(grads["D"][0], fx["step0"], "D", 0, bound_d)
    # state after step 0 against the reference's (parameters: +-lr per element whose gradient is rounding noise; u/v: tight)
    for mod, key in ((G, "G_sd1"), (D, "D_sd1")):
        sd = mod.state_dict()
        for k, v in fx[key].items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-4, atol=1e-6, msg=lambda mm, k=k: f"{k}: {mm}")
            else:
                assert float((sd[k].cpu() - v).abs().max()) <= 2.2 * cfg["lr"], k
    s = fx["step1"]
    gen, losses = run_step(G, D, gopt, s)
    torch.testing.assert_close(gen.detach().cpu(), s["gen"], rtol=0, atol=5e-3)
    for got, key in zip(losses, ("d_loss", "g_loss", "perc", "style", "cx")):
        torch.testing.assert_close(got.detach().cpu(), s[key], rtol=2e-2, atol=1e-9, msg=lambda mm, key=key: f"{key} step 1 (own state): {mm}")
    # an element whose gradient is ~0 can move by +-lr per step in either run: hard bound 2 runs x 2 steps x lr (+10 %)
    for mod, key in ((G, "G_sd2"), (D, "D_sd2")):
        sd = mod.state_dict()
        for k, v in fx[key].items():
            d = (sd[k].cpu() - v).abs()
            assert d.max() <= 4.4 * cfg["lr"], f"{key} {k}: max diff {d.max():.3e}"

    # ---- step 1 restarted from the reference's state at the start of that step
    f1 = dict(fx)
    f1["G_sd0"], f1["D_sd0"] = fx["G_sd1"], fx["D_sd1"]
    G, D, gopt, optG, optD = _tiny_models(f1, dev)
    grads = {"G": [], "D": []}
    _spy(optG, list(G.named_parameters()), grads["G"])
    _spy(optD, list(D.named_parameters()), grads["D"])
    gen, losses = run_step(G, D, gopt, s)
    check_outputs(gen, losses, s, 1)
    _check_grads_fp64(grads["G"][0], s, "G", 1, bound_g)
    _check_grads_fp64(grads["D"][0], s, "D", 1, bound_d)
    # SpectralNorm state after this step (2 G forwards / 6 D forwards in total) against the reference's end state
    for mod, key in ((G, "G_sd2"), (D, "D_sd2")):
        sd = mod.state_dict()
        for k, v in fx[key].items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                torch.testing.assert_close(sd[k].cpu(), v, rtol=1e-4, atol=1e-6, msg=lambda mm, k=k: f"{k}: {mm}")


def test_forward_matches_oracle_at_moderate_size(dev):
    """real channel widths (ngf 32 / img_f 128 / decoder 256), 64x64 input: 4096-token decoder attention"""
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd.modules.model import ReferenceFill
    from oracle import picnet_cpu as O  # checker

    torch.manual_seed(1)
    enc = dict(type="pluralistic", ngf=32, z_nc=128, img_f=128, layers=5, norm="none", activation="LeakyReLU", L=6)
    dec = dict(ngf=32, z_nc=256, img_f=256, layers=5, norm="instance", activation="LeakyReLU", L=0)
    G = ReferenceFill(None, dict(enc), dict(dec), use_att=True, out_size=(64, 64))
    with torch.no_grad():
        G.decoder.attn1.gamma.fill_(0.5)
    P = O.prepare_params(G.state_dict())
    G = G.to(dev)
    src, ref, gt, mask, eps_p, eps_q = O.synthetic_batch(2, 64, seed=5, feat_hw=8, z_nc=128)
    with torch.no_grad():
        want = O.reference_fill_forward(P, src, ref, O.binarise_mask(mask), eps_p, eps_q, out_size=(64, 64))
        got = G(src.to(dev), ref.to(dev), src_mask=FF.binarise_mask(mask.to(dev)), eps=(eps_p.to(dev), eps_q.to(dev)))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-3, atol=1e-4)
    # state after the forward (u/v advanced once) must agree too
    sd = G.state_dict()
    for k in ("decoder.decoder4.conv2.module.weight_u", "src_encoder.prior.conv1.module.weight_v"):
        torch.testing.assert_close(sd[k].cpu(), P[k], rtol=1e-4, atol=1e-6)


def test_early_discriminator_schedule_equals_reference_order(dev, golden):
    """The data-parallel schedule (D loss forward/backward before the generator backward, so that the D all-reduce can
    overlap the VGG dgrad) must not change any value: same losses, same G / D gradients at step 0 (incl. the SpectralNorm
    u/v sequence gen -> gt -> gen.detach()), same losses at step 1."""
    from face_mask_inpaint_amd import functional as FF

    fx = golden("picnet_train_tiny.pt")
    runs = {}
    for early in (False, True):
        G, D, gopt, optG, optD = _tiny_models(fx, dev)
        gopt.early_d = early
        cap = {}

        def spy(opt, named, key):
            orig = opt.step

            def step(closure=None):
                cap.setdefault(key, {n: p.grad.detach().clone() for n, p in named if p.grad is not None})
                return orig(closure)

            opt.step = step

        spy(optG, list(G.named_parameters()), "G")
        spy(optD, list(D.named_parameters()), "D")
        losses = []
        for s in (fx["step0"], fx["step1"]):
            m = FF.binarise_mask(s["mask"].to(dev))
            gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
            losses.append([float(v) for v in gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)])
        uv = {k: v.clone() for k, v in D.state_dict().items() if k.endswith("weight_u")}
        runs[early] = (cap, losses, uv)
    (ca, la, ua), (cb, lb, ub) = runs[False], runs[True]
    for key in ("G", "D"):
        assert ca[key].keys() == cb[key].keys()
        for n, g in ca[key].items():
            # the two runs differ only by the summation order of the fp32 atomics (split reductions, weight gradients), which the
            # tiny model's conditioning amplifies; a wrong schedule would be off by O(1)
            err, lim = float((cb[key][n] - g).abs().max()), 3e-4 * float(g.abs().max()) + 3e-9
            assert err <= lim, f"{key}.{n}: {err:.3e} > {lim:.3e}"
    for a, b in zip(la[0], lb[0]):
        assert abs(a - b) <= 1e-6 * abs(a) + 1e-12
    for a, b in zip(la[1], lb[1]):
        assert abs(a - b) <= 1e-3 * abs(a) + 1e-10
    for k in ua:
        torch.testing.assert_close(ua[k], ub[k], rtol=1e-3, atol=1e-5)


def test_generator_gradients_without_contextual_term(dev, golden):
    """G_loss minus the contextual term: GAN + L1 + perceptual + style, backward through the whole generator; and the contextual
    term alone on d cx / d gen.  Adjudicated by the CPU oracle evaluated in FLOAT64 (the same restatement the golden vectors pin in
    fp32; its float64 form is pinned to the reference's float64 gradients in tests/test_oracle_golden.py): the HIP error against
    float64 may not exceed twice the error of the oracle's own fp32 evaluation, per tensor where the loss is smooth (d cx / d gen)
    and network-wide where LeakyReLU / ReLU kink flips decide (see _check_grads_fp64)."""
    import torch.nn.functional as F

    from face_mask_inpaint_amd import functional as FF
    from oracle import picnet_cpu as O  # checker

    fx = golden("picnet_train_tiny.pt")
    cfg, s = fx["config"], fx["step0"]
    G, D, gopt, optG, optD = _tiny_models(fx, dev)
    kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"],
              dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)
    sd_g = {k: v.cpu() for k, v in G.state_dict().items()}
    sd_d = {k: v.cpu() for k, v in D.state_dict().items()}
    names = [n for n, _ in G.named_parameters()]
    ora = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        PG = O.prepare_params(sd_g, dtype=dt)
        PD = O.prepare_params(sd_d, frozen=True, dtype=dt)
        PV = O.prepare_params(fx["V_sd"], frozen=True, dtype=dt)
        mask = O.binarise_mask(s["mask"]).to(dt)
        src, gt, ref = s["src"].to(dt), s["gt"].to(dt), s["ref"].to(dt)
        ogen = O.reference_fill_forward(PG, src, ref, mask, s["eps_p"].to(dt), s["eps_q"].to(dt), **kw)
        ototal = (O.lsgan(O.res_discriminator(PD, "", ogen, cfg["disc_layers"]), True) * O.LAMBDA_G + F.l1_loss(ogen, gt)
                  + O.vgg_loss(PV, "", ogen, gt, "perceptual") * O.LAMBDA_PERC
                  + O.vgg_loss(PV, "", ogen * (1 - mask).unsqueeze(1), src, "style") * O.LAMBDA_STYLE)
        ototal.backward()
        og = s["gen"].to(dt).clone().requires_grad_(True)  # common evaluation point of d cx / d gen: the reference's image
        (ocx_g,) = torch.autograd.grad(O.vgg_loss(PV, "", og * mask.unsqueeze(1), ref * mask.unsqueeze(1), "contextual"), og)
        ora[tag] = (float(ototal), {n: PG[n].grad.double() for n in names if PG[n].grad is not None}, ocx_g.double())

    m = FF.binarise_mask(s["mask"].to(dev))
    src, gt, ref = s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev)
    gen = G(src, ref, src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    for p in D.parameters():
        p.requires_grad_(False)
    perc, sty = gopt.vgg_loss.forward_multi([(gen, gt, "perceptual"), (gopt._masked(gen, m, True), src, "style")])
    total = gopt.generator_loss(D, gt, gen, freeze=False) + perc * gopt.lambda_perc + sty * gopt.lambda_style
    assert abs(float(total) / ora["f64"][0] - 1) <= 1e-5
    total.backward()
    net_bound = 2 * _reference_fp32_error(fx, "G")
    g64, g32 = ora["f64"][1], ora["f32"][1]
    worst = []
    for n, p in G.named_parameters():
        if n not in g64:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        mx = float(g64[n].abs().max())
        if mx <= 1e-12:
            continue
        err = float((p.grad.cpu().double() - g64[n]).abs().max()) / mx
        ref_err = float((g32[n] - g64[n]).abs().max()) / mx
        assert err <= net_bound, f"{n}: {err:.3e} of max|g| > {net_bound:.3e} (oracle fp32: {ref_err:.3e})"
        worst.append((err, ref_err, n))
    worst.sort(reverse=True)
    print("worst errors against float64 (HIP, oracle fp32):", [("%.2e" % a, "%.2e" % b, c) for a, b, c in worst[:6]])
    g2 = s["gen"].to(dev).clone().requires_grad_(True)
    (cx_g,) = torch.autograd.grad(gopt.contextual_loss(g2, ref, m), g2)
    c64, c32 = ora["f64"][2], ora["f32"][2]
    e_hip = float((cx_g.cpu().double() - c64).norm() / c64.norm())
    e_o32 = float((c32 - c64).norm() / c64.norm())
    assert e_hip <= 2 * e_o32 + 1e-5, f"d cx / d gen: relative L2 error {e_hip:.3e} against float64 > 2 x the fp32 oracle's {e_o32:.3e}"


def test_patch_discriminator_against_reference_golden(dev, golden):
    """define_d(model_type='PatchDis') (network.py:373-430): 4x4 stride-2 / stride-1 SpectralNorm convs"""
    from face_mask_inpaint_amd.modules.pluralistic_model import network

    fx = golden("picnet_patchdis.pt")
    d = network.define_d(ndf=8, img_f=32, layers=3, norm="none", activation="LeakyReLU", model_type="PatchDis")
    assert set(d.state_dict()) == set(fx["sd0"])
    d.load_state_dict(fx["sd0"])
    d.to(dev)
    x = fx["x"].to(dev).requires_grad_(True)
    y = d(x)
    torch.testing.assert_close(y.detach().cpu(), fx["out"], rtol=1e-4, atol=1e-5)
    y.backward(fx["gout"].to(dev))
    torch.testing.assert_close(x.grad.cpu(), fx["gx"], rtol=2e-4, atol=1e-5)
    P = dict(d.named_parameters())
    for n, g in fx["gparams"].items():
        torch.testing.assert_close(P[n].grad.cpu(), g, rtol=2e-4, atol=2e-5, msg=lambda m, n=n: f"{n}: {m}")
    sd1 = d.state_dict()
    for k, v in fx["sd1"].items():
        if k.endswith("weight_u") or k.endswith("weight_v"):
            torch.testing.assert_close(sd1[k].cpu(), v, rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        network.define_d(model_type="nope")


def test_training_steps_do_not_accumulate_device_tensors(dev, golden):
    """every step must release the previous step's graph: the count of live device tensors is the same after step 3 and step 7
    (a reference cycle node -> ctx attribute -> output tensor -> grad_fn once kept one discriminator graph alive per step)"""
    import gc

    from face_mask_inpaint_amd import functional as FF

    fx = golden("picnet_train_tiny.pt")
    G, D, gopt, _, _ = _tiny_models(fx, dev)
    s = fx["step0"]
    m = FF.binarise_mask(s["mask"].to(dev))
    src, ref, gt = s["src"].to(dev), s["ref"].to(dev), s["gt"].to(dev)
    eps = (s["eps_p"].to(dev), s["eps_q"].to(dev))

    def live():
        gc.collect()
        return sum(1 for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda)

    counts = []
    for i in range(8):
        gen = G(src, ref, src_mask=m, eps=eps)
        out = gopt(D, src, gt, ref, gen, m)
        del gen, out
        if i in (3, 7):
            torch.cuda.synchronize()
            counts.append(live())
    assert counts[1] <= counts[0] + 2, counts
