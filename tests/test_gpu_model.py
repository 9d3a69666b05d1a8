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
    _run_block(fx["auto_attn"], bf.Auto_Attn(16, None), dev)
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


def test_two_training_steps_against_reference_golden(dev, golden):
    """ReferenceFill.forward + GANOptimizer.__call__ for two consecutive steps.

    Step 0 is checked against the golden vectors of the imported reference: generated image, all five losses and
    every parameter gradient (captured just before each optimiser step).
    Adam's first updates are lr * g / (|g| + 1e-8): for parameters whose gradient is at the 1e-8 level (tiny
    fixture, lr = 1e-3) rounding noise in g is amplified into the 1e-5..1e-4 range of the parameters, so step 1 is
    checked two ways: (a) strictly against the CPU oracle re-started from THIS run's post-step-0 state (isolates the
    second forward/backward incl. the evolved SpectralNorm u/v), (b) loosely against the golden end state."""
    from face_mask_inpaint_amd import functional as FF
    from oracle import picnet_cpu as O  # checker

    fx = golden("picnet_train_tiny.pt")
    cfg = fx["config"]
    G, D, gopt, optG, optD = _tiny_models(fx, dev)
    grads = {"G": [], "D": []}

    def spy(opt, names_params, key):
        orig = opt.step

        def step(closure=None):
            grads[key].append({n: p.grad.detach().cpu().clone() for n, p in names_params if p.grad is not None})
            return orig(closure)

        opt.step = step

    spy(optG, list(G.named_parameters()), "G")
    spy(optD, list(D.named_parameters()), "D")

    def check_grads(got, ref, key, step):
        for n, g in ref.items():
            assert n in got, f"missing grad {key}.{n}"
            # relative to the tensor's largest entry; the 1e-7 floor covers gradients that are analytically zero
            # (a conv bias in front of InstanceNorm) and hold only rounding noise on both sides
            # Bias gradients are sums over all pixels of signed terms (cancellation): their error is bounded
            # relative to sum|terms|, not to the result, hence the wider absolute floor for 1-D tensors.
            # End-to-end gradients of this tiny random-init network are ill-conditioned with respect to forward rounding: a
            # 1e-7 relative perturbation of ONE early convolution output moves the generator's parameter gradients by up to
            # 1.2e-3 of their largest entry (tools/bench_tools/perturb.py, measured with one and the same library build), i.e. an
            # amplification of ~1e4; every GEMM-family kernel rounds differently from the CPU reference at the 1e-7 level, so
            # ~30 such contributions bound the generator's gradients at 4e-2.  Kernel- and block-level tests hold 1e-5..2e-4.
            err = (got[n] - g).abs().max()
            rel = 4e-2 if key == "G" else 5e-3
            lim = (max(rel, 1e-2) * g.abs().max() + 1e-6) if g.ndim == 1 else (rel * g.abs().max() + 1e-7)
            assert err <= lim, f"{key} grad {n} step {step}: max error {err:.3e} > {lim:.3e}"

    def run_step(s):
        m = FF.binarise_mask(s["mask"].to(dev))
        assert torch.equal(m.cpu(), (s["mask"] > 0).float())
        gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
        return gen, gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)

    # ---- step 0 against the reference's golden vectors
    s = fx["step0"]
    gen, (d_loss, g_loss, perc, sty, cx) = run_step(s)
    torch.testing.assert_close(gen.detach().cpu(), s["gen"], rtol=1e-3, atol=1e-5)
    for got, key in ((g_loss, "g_loss"), (d_loss, "d_loss"), (perc, "perc"), (sty, "style"), (cx, "cx")):
        torch.testing.assert_close(got.detach().cpu(), s[key], rtol=1e-3, atol=1e-9, msg=lambda mm, key=key: f"{key} step 0: {mm}")
    check_grads(grads["G"][0], s["G_grads"], "G", 0)
    check_grads(grads["D"][0], s["D_grads"], "D", 0)

    # ---- step 1 against the oracle restarted from this run's state
    PG = O.prepare_params({k: v.cpu() for k, v in G.state_dict().items()})
    PD = O.prepare_params({k: v.cpu() for k, v in D.state_dict().items()})
    PV = O.prepare_params(fx["V_sd"], frozen=True)
    s = fx["step1"]
    mask = O.binarise_mask(s["mask"])
    kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"],
              dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)
    ogen = O.reference_fill_forward(PG, s["src"], s["ref"], mask, s["eps_p"], s["eps_q"], **kw)
    import torch.nn.functional as F

    og = O.lsgan(O.res_discriminator(PD, "", ogen, cfg["disc_layers"]), True) * O.LAMBDA_G + F.l1_loss(ogen, s["gt"])
    operc = O.vgg_loss(PV, "", ogen, s["gt"], "perceptual") * O.LAMBDA_PERC
    osty = O.vgg_loss(PV, "", ogen * (1 - mask).unsqueeze(1), s["src"], "style") * O.LAMBDA_STYLE
    ocx = O.vgg_loss(PV, "", ogen * mask.unsqueeze(1), s["ref"] * mask.unsqueeze(1), "contextual") * O.LAMBDA_CX
    og_total = og + operc + osty + ocx
    og_total.backward()
    ograds_g = {n: PG[n].grad.clone() for n, _ in G.named_parameters() if PG[n].grad is not None}
    od = (O.lsgan(O.res_discriminator(PD, "", s["gt"], cfg["disc_layers"]), True)
          + O.lsgan(O.res_discriminator(PD, "", ogen.detach(), cfg["disc_layers"]), False)) * 0.5
    for t in PD.values():
        t.grad = None
    od.backward()
    ograds_d = {n: PD[n].grad.clone() for n, _ in D.named_parameters() if PD[n].grad is not None}

    gen, (d_loss, g_loss, perc, sty, cx) = run_step(s)
    torch.testing.assert_close(gen.detach().cpu(), ogen.detach(), rtol=1e-3, atol=1e-5)
    for got, want, key in ((g_loss, og_total, "g_loss"), (d_loss, od, "d_loss"), (perc, operc, "perc"), (sty, osty, "style"), (cx, ocx, "cx")):
        torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-3, atol=1e-9, msg=lambda mm, key=key: f"{key} step 1: {mm}")
    check_grads(grads["G"][1], ograds_g, "G", 1)
    check_grads(grads["D"][1], ograds_d, "D", 1)
    # SpectralNorm state after 2 G forwards / 6 D forwards agrees with the oracle's (which started from ours after step 0)
    for mod, P in ((G, PG), (D, PD)):
        for k, v in mod.state_dict().items():
            alias = ".shortcut." in "." + k or re.search(r"(^|\.)model\.\d+\.module\.", k)
            if (k.endswith("weight_u") or k.endswith("weight_v")) and not alias and "attn" not in k.split(".model.")[0][-6:]:
                torch.testing.assert_close(v.cpu(), P[k], rtol=1e-4, atol=1e-6, msg=lambda mm, k=k: f"{k}: {mm}")

    # ---- (b) end state against the golden: the generated image of step 1 and the parameters after two Adam steps
    torch.testing.assert_close(gen.detach().cpu(), s["gen"], rtol=0, atol=5e-3)
    # an element whose gradient is ~0 can move by +-lr per step in either run: hard bound 2 runs x 2 steps x lr (+10 %);
    # the bulk agrees far better (typically > 98 % of the elements within 2e-4), which is informational only
    for mod, key in ((G, "G_sd2"), (D, "D_sd2")):
        sd = mod.state_dict()
        for k, v in fx[key].items():
            d = (sd[k].cpu() - v).abs()
            assert d.max() <= 4.4 * cfg["lr"], f"{key} {k}: max diff {d.max():.3e}"


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
    """G_loss minus the (ill-conditioned) contextual term: GAN + L1 + perceptual + style, backward through the whole
    generator, against the CPU oracle on the same parameters (bounds: see the conditioning note in the test above).
    The contextual term alone is compared on d cx / d gen with the bound its conditioning allows."""
    import torch.nn.functional as F

    from face_mask_inpaint_amd import functional as FF
    from oracle import picnet_cpu as O  # checker

    fx = golden("picnet_train_tiny.pt")
    cfg, s = fx["config"], fx["step0"]
    G, D, gopt, optG, optD = _tiny_models(fx, dev)
    PG = O.prepare_params({k: v.cpu() for k, v in G.state_dict().items()})
    PD = O.prepare_params({k: v.cpu() for k, v in D.state_dict().items()}, frozen=True)
    PV = O.prepare_params(fx["V_sd"], frozen=True)
    mask = O.binarise_mask(s["mask"])
    kw = dict(enc_layers=cfg["enc_layers"], enc_L=cfg["enc_L"], enc_z_nc=cfg["enc_z_nc"], dec_layers=cfg["dec_layers"],
              dec_L=cfg["dec_L"], out_size=(cfg["out_size"],) * 2)
    ogen = O.reference_fill_forward(PG, s["src"], s["ref"], mask, s["eps_p"], s["eps_q"], **kw)
    ototal = (O.lsgan(O.res_discriminator(PD, "", ogen, cfg["disc_layers"]), True) * O.LAMBDA_G + F.l1_loss(ogen, s["gt"])
              + O.vgg_loss(PV, "", ogen, s["gt"], "perceptual") * O.LAMBDA_PERC
              + O.vgg_loss(PV, "", ogen * (1 - mask).unsqueeze(1), s["src"], "style") * O.LAMBDA_STYLE)
    ototal.backward()
    og = ogen.detach().clone().requires_grad_(True)
    (ocx_g,) = torch.autograd.grad(O.vgg_loss(PV, "", og * mask.unsqueeze(1), s["ref"] * mask.unsqueeze(1), "contextual"), og)

    m = FF.binarise_mask(s["mask"].to(dev))
    src, gt, ref = s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev)
    gen = G(src, ref, src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    for p in D.parameters():
        p.requires_grad_(False)
    perc, sty = gopt.vgg_loss.forward_multi([(gen, gt, "perceptual"), (gopt._masked(gen, m, True), src, "style")])
    total = gopt.generator_loss(D, gt, gen, freeze=False) + perc * gopt.lambda_perc + sty * gopt.lambda_style
    torch.testing.assert_close(total.detach().cpu(), ototal.detach(), rtol=1e-4, atol=1e-9)
    total.backward()
    worst = []
    for n, p in G.named_parameters():
        want = PG[n].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        # per tensor: largest deviation within 4e-2 of the largest entry and relative L2 error within 2e-2 -- the conditioning
        # bound explained in test_two_training_steps_against_reference_golden (measured: tools/bench_tools/perturb.py)
        err = float((p.grad.cpu() - want).abs().max())
        lim = 4e-2 * float(want.abs().max()) + (1e-6 if want.ndim == 1 else 1e-7)
        assert err <= lim, f"{n}: {err:.3e} > {lim:.3e}"
        if want.ndim > 1:
            l2 = float((p.grad.cpu() - want).norm() / (want.norm() + 1e-30))
            worst.append((l2, n))
    worst.sort(reverse=True)
    print("worst relative L2 errors:", [("%.2e" % a, b) for a, b in worst[:6]])
    for l2, n in worst:
            assert l2 <= 2e-2, f"{n}: relative L2 error {l2:.3e}"
    g2 = gen.detach().clone().requires_grad_(True)
    (cx_g,) = torch.autograd.grad(gopt.contextual_loss(g2, ref, m), g2)
    rel_l2 = float((cx_g.cpu() - ocx_g).norm() / ocx_g.norm())
    assert rel_l2 <= 5e-2, f"d cx / d gen: relative L2 error {rel_l2:.3e}"


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
