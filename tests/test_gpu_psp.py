"""GPU: the pSp encoder side through the reference-named modules, against the golden
vectors produced by the reference's own classes (tests/golden/psp_ops.pt) and, for the full-width pSp forward, against the
CPU oracle.  fp32 activations: 1e-3 relative (BASELINE.json north_star), gradients scaled by their largest entry."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(got, want, rel=1e-3, what=""):
    want = want.to(torch.float32)
    tol = rel * float(want.abs().max()) + 1e-7
    err = float((got.detach().cpu().float() - want).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e}"


def _run_block(m, fx, dev, rel=1e-3, only=None):
    m.load_state_dict(fx["sd"])
    m.to(dev).train()
    x = fx["x"].to(dev).requires_grad_(True)
    y = m(x)
    _close(y, fx["out"], rel, "out")
    y.backward(fx["gout"].to(dev))
    _close(x.grad, fx["gx"], rel, "gx")
    P = dict(m.named_parameters())
    for n, g in fx["gparams"].items():
        _close(P[n].grad, g, 2e-3, n)
    sd = m.state_dict()
    for k, v in fx["stats_after"].items():
        if v.is_floating_point():
            _close(sd[k], v, 1e-4, k)
        else:
            assert int(sd[k]) == int(v), k
    m.eval()
    with torch.no_grad():
        _close(m(fx["x"].to(dev)), fx["out_eval"], rel, "eval")


@pytest.mark.parametrize("name,args,se", [("ir_se_conv_s2", (16, 32, 2), True), ("ir_se_pool_s1", (32, 32, 1), True),
                                          ("ir_se_pool_s2", (32, 32, 2), True), ("ir_conv_s2", (8, 24, 2), False)])
def test_bottlenecks_against_reference_golden(dev, golden, name, args, se):
    from face_mask_inpaint_amd.modules.psp.encoders import helpers as H

    _run_block((H.bottleneck_IR_SE if se else H.bottleneck_IR)(*args), golden("psp_ops.pt")[name], dev)


def test_style_block_against_reference_golden(dev, golden):
    from face_mask_inpaint_amd.modules.psp.encoders.psp_encoders import GradualStyleBlock

    fx = dict(golden("psp_ops.pt")["style_block"])
    _run_block(GradualStyleBlock(16, 16, 8), fx, dev)


def test_encoder_against_reference_golden(dev, golden):
    from face_mask_inpaint_amd.modules.psp.encoders.psp_encoders import GradualStyleEncoder

    fx = golden("psp_ops.pt")["encoder"]
    opts = types.SimpleNamespace(n_styles=fx["n_styles"], use_attention=True)
    enc = GradualStyleEncoder(50, "ir_se", opts, _widths=tuple(fx["widths"]), _spatial=tuple(fx["spatial"]))
    enc.load_state_dict(fx["sd"])
    enc.to(dev).train()
    x, ref, mask = fx["x"].to(dev).requires_grad_(True), fx["ref"].to(dev).requires_grad_(True), fx["mask"].to(dev)
    out = enc(x, ref=ref, mask=mask)
    assert out.shape == fx["out"].shape
    _close(out, fx["out"], 1e-3, "codes")
    out.backward(fx["gout"].to(dev))
    _close(x.grad, fx["gx"], 2e-3, "gx")
    _close(ref.grad, fx["gref"], 2e-3, "gref")
    P = dict(enc.named_parameters())
    for n, g in fx["gparams"].items():
        _close(P[n].grad, g, 5e-3, n)
    sd = enc.state_dict()
    for k, v in fx["stats_after"].items():
        if v.is_floating_point():
            _close(sd[k], v, 1e-3, k)
        else:
            assert int(sd[k]) == int(v), k
    enc.eval()
    with torch.no_grad():
        _close(enc(fx["x"].to(dev), ref=fx["ref"].to(dev), mask=mask), fx["out_eval"], 1e-3, "eval")
        _close(enc(fx["x"].to(dev)), fx["out_eval_noref"], 1e-3, "eval noref")
        enc.use_attention = False
        _close(enc(fx["x"].to(dev), ref=fx["ref"].to(dev), mask=mask), fx["out_eval_noatt"], 1e-3, "eval noatt")


def test_psp_loss_against_reference_golden(dev, golden):
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss

    fx = golden("psp_ops.pt")["psp_loss"]
    a = types.SimpleNamespace(**fx["args"])
    from face_mask_inpaint_amd.modules.loss import VGGLoss

    crit = pSpLoss(a)
    crit.vgg_loss = VGGLoss(width_div=8)  # the fixture's stand-in VGG (torchvision weights are not obtainable offline)
    crit.vgg_loss.load_state_dict(fx["vgg"])
    crit.to(dev)
    yh, lat = fx["y_hat"].to(dev).requires_grad_(True), fx["latent"].to(dev).requires_grad_(True)
    loss, ld, _ = crit(fx["x"].to(dev), fx["y"].to(dev), yh, lat, latent_avg=fx["latent_avg"], ref=fx["ref"].to(dev), mask=fx["mask"].to(dev))
    _close(loss, fx["loss"], 1e-5, "loss")
    for k, v in fx["loss_dict"].items():
        assert abs(ld[k] - float(v)) <= 1e-3 * abs(float(v)) + 1e-7, (k, ld[k], float(v))
    loss.backward()
    _close(yh.grad, fx["gy_hat"], 1e-4, "gy_hat")
    _close(lat.grad, fx["glatent"], 1e-4, "glatent")
    loss2, ld2, _ = crit(fx["x"].to(dev), fx["y"].to(dev), fx["y_hat"].to(dev), fx["latent"].to(dev))
    _close(loss2, fx["loss_nomask"], 1e-5, "loss nomask")
    # the LPIPS / ID terms are built on demand (test_lpips_id_and_full_psp_loss_against_reference)
    assert hasattr(pSpLoss(types.SimpleNamespace(**{**fx["args"], "lpips_lambda": 0.8})), "lpips_loss") and not hasattr(crit, "lpips_loss")


@pytest.mark.parametrize("reproducible", [False, True])
def test_psp_whole_train_against_reference(dev, golden, reproducible):
    """reproducible = the library's deterministic mode: the gradients are then a FIXED function of the inputs (two runs are bit-identical,
    tools/bench_tools/psp_det_probe.py), so the bounds below that absorb run-to-run spread in the default mode are replaced by the
    measured values with a small margin: parameter-gradient error against float64 median 5.2e-4 / p90 1.7e-3 / worst 5.3e-2 (the
    reference's own fp32 run: 2.2e-4 / 9.7e-4 / 2.1e-2 -- training-mode BatchNorm at batch 2 is chaotic at this level, the default mode
    lands between 1.6e-4 and 4.5e-4 from run to run), eval mode median 6.1e-5 / p90 1.7e-4, scalar noise weights 4.8e-2.

    the WHOLE pSp at full widths (IR-SE50 GradualStyleEncoder with attention on src + ref, latent_avg, 256^2 StyleGAN2 decoder:
    BASELINE configs[2] shapes at batch 2) in TRAINING mode, forward + backward, against the imported reference
    (tests/golden/psp_whole.pt, oracle/gen_golden.py:psp_whole_fixture; parameters from oracle/seeded.py on both sides): image and
    W+ codes at 1e-3, the input gradients and EVERY parameter gradient adjudicated by the reference's float64 run, BatchNorm
    running statistics after the step, then the eval-mode image"""
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from oracle.seeded import check_adjudicated, check_digest, digest_error, seeded_fill_, seeded_tensor  # checker

    fx = golden("psp_whole.pt")
    cfg = fx["config"]
    opts = types.SimpleNamespace(output_size=cfg["output_size"], encoder_type="GradualStyleEncoder", use_attention=True, train_decoder=True,
                                 start_from_latent_avg=True, learn_in_w=False, pt_ckpt_path=None, stylegan_weights=None)
    net = pSp(opts)
    seeded_fill_(net, cfg["seed"])
    net.latent_avg = seeded_tensor((opts.n_styles, 512), cfg["latent_avg_seed"], 0.5)
    net.to(dev).train()
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["x_seed"])) * 2 - 1).to(dev).requires_grad_(True)
    ref = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(cfg["ref_seed"])) * 2 - 1).to(dev).requires_grad_(True)
    mask = torch.zeros(2, 256, 256)
    for i, (a, b, c, d) in enumerate(cfg["rects"]):
        mask[i, a:b, c:d] = 1
    mask = mask.to(dev)
    from face_mask_inpaint_amd import functional as FF

    ctx = FF.deterministic(reproducible)
    ctx.__enter__()
    request_cleanup = ctx  # left by the finally below
    try:
        _psp_whole_body(net, x, ref, mask, fx, cfg, dev, reproducible, check_adjudicated, check_digest, digest_error, seeded_tensor)
    finally:
        request_cleanup.__exit__(None, None, None)


def _psp_whole_body(net, x, ref, mask, fx, cfg, dev, reproducible, check_adjudicated, check_digest, digest_error, seeded_tensor):
    img, lat = net(x, ref=ref, src_mask=mask, resize=True, randomize_noise=False, return_latents=True)
    _close(lat, fx["latent"], 1e-3, "W+ codes")
    _close(img, fx["image"], 1e-3, "image")
    ((img * seeded_tensor(img.shape, cfg["cot_seeds"][0]).to(dev)).sum() / 256.0 + (lat * seeded_tensor(lat.shape, cfg["cot_seeds"][1]).to(dev)).sum()).backward()
    check_adjudicated({"gx": x.grad, "gref": ref.grad}, {"gx": fx["gx"], "gref": fx["gref"]}, {"gx": fx["gx64"], "gref": fx["gref64"]},
                      floor=5e-3, what="pSp input gradients (HIP)")
    P = dict(net.named_parameters())
    # measured over four runs: median 3.9e-4 .. 4.5e-4 / p90 1.8e-3 .. 2.2e-3 (run-to-run: fp32 atomics) against the reference's
    # 2.2e-4 / 9.7e-4; the eval-mode comparison below is the strict one -- the IR-SE50 convolutions accumulate up to 4608
    # products sequentially in one fp32 MFMA accumulator where oneDNN adds blocked partial sums, so the forward rounding that feeds
    # the kink flips is ~1.7x the CPU's (the decoder, by contrast, is 30x CLOSER to float64 than the reference: test_gpu_stylegan2_ops)
    check_adjudicated({n: P[n].grad for n in fx["gparams64"]}, fx["gparams"], fx["gparams64"], what="pSp parameters (HIP)",
                      med_factor=2.6 if reproducible else 3.5, p90_factor=2.0 if reproducible else 4.5)
    assert sorted(n for n, p in P.items() if p.grad is None) == fx["no_grad"]
    sd = net.state_dict()
    for k, v in fx["stats_after"].items():
        if v.is_floating_point():
            _close(sd[k], v, 1e-4, k)
        else:
            assert int(sd[k]) == int(v), k
    # eval mode (frozen BatchNorm statistics): the well-conditioned form of the same forward + backward -- strict bounds
    net.eval()
    net.zero_grad()
    xe, re = x.detach().clone().requires_grad_(True), ref.detach().clone().requires_grad_(True)
    img_e, lat_e = net(xe, ref=re, src_mask=mask, resize=True, randomize_noise=False, return_latents=True)
    check_digest(img_e, fx["image_eval"], 1e-3, "eval image")
    _close(lat_e, fx["eval"]["latent"], 1e-4, "eval W+ codes")
    ((img_e * seeded_tensor(img_e.shape, cfg["cot_seeds"][0]).to(dev)).sum() / 256.0 + (lat_e * seeded_tensor(lat_e.shape, cfg["cot_seeds"][1]).to(dev)).sum()).backward()
    errs = [(digest_error(xe.grad, fx["eval"]["gx"]), "gx"), (digest_error(re.grad, fx["eval"]["gref"]), "gref")]
    scalars = []
    for n, d in fx["eval"]["gparams"].items():
        if float(d["max"]) > 1e-20:
            (scalars if P[n].numel() == 1 else errs).append((digest_error(P[n].grad, d), n))
    errs.sort()
    scalars.sort()
    print("pSp eval-mode gradient errors vs the reference (fp32): median %.2e p90 %.2e worst %.2e (%s); scalar noise weights worst %.2e" % (
        errs[len(errs) // 2][0], errs[int(0.9 * len(errs))][0], errs[-1][0], errs[-1][1], scalars[-1][0]))
    assert errs[len(errs) // 2][0] <= 2e-4 and errs[int(0.9 * len(errs))][0] <= 1e-3 and errs[-1][0] <= 2e-2, errs[-4:]
    # the 13 one-element noise-weight gradients are sums of 10^5 .. 10^6 signed terms g * noise that nearly cancel: the reference's
    # own fp32 run is 2e-2 from float64 on them, and the HIP value moves by a few 1e-2 from run to run (fp32 atomics)
    assert scalars[-1][0] <= (0.06 if reproducible else 0.2), scalars[-3:]
    if reproducible:
        assert errs[len(errs) // 2][0] <= 8e-5 and errs[int(0.9 * len(errs))][0] <= 2.2e-4 and errs[-1][0] <= 6e-3, errs[-4:]


def test_lpips_id_and_full_psp_loss_against_reference(dev, golden):
    """LPIPS(alex), IDLoss (ArcFace IR-SE50, eval) and pSpLoss.__call__ with every lambda on -- scripts/train_psp.sh's loss -- against
    the reference's own forward code on seeded parameters (tests/golden/psp_criteria.pt): values 1e-3, gradients w.r.t. y_hat and the
    latent codes"""
    from oracle.seeded import check_digest, criteria_inputs  # checker
    from test_oracle_criteria import criterion

    fx = golden("psp_criteria.pt")
    crit = criterion(fx).to(dev)
    x, y, rf, yh, mask = (t.to(dev) for t in criteria_inputs(fx["seeds"]["inputs"]))
    yh.requires_grad_(True)
    v = crit.lpips_loss(yh, y)
    assert abs(float(v) / float(fx["lpips"]["out"]) - 1) <= 1e-3
    v.backward()
    check_digest(yh.grad, fx["lpips"]["gy_hat"], 2e-3, "d lpips / d y_hat")
    yh.grad = None
    l, imp, logs = crit.id_loss(yh, y, x)
    assert abs(float(l) / float(fx["id"]["loss"]) - 1) <= 1e-3 and abs(imp - float(fx["id"]["improve"])) <= 1e-3
    torch.testing.assert_close(torch.tensor([[d["diff_target"], d["diff_input"], d["diff_views"]] for d in logs]), fx["id"]["logs"], rtol=1e-3, atol=1e-3)
    l.backward()
    check_digest(yh.grad, fx["id"]["gy_hat"], 5e-3, "d id / d y_hat")
    yh.grad = None
    f = fx["psp_loss_full"]
    lat = f["latent"].to(dev).requires_grad_(True)
    loss, ld, id_logs = crit(x, y, yh, lat, latent_avg=f["latent_avg"].to(dev), ref=rf, mask=mask)
    assert abs(float(loss) / float(f["loss"]) - 1) <= 1e-3
    for k, want in f["loss_dict"].items():
        assert abs(ld[k] - float(want)) <= 1e-3 * abs(float(want)) + 1e-6, (k, ld[k], float(want))
    loss.backward()
    check_digest(yh.grad, f["gy_hat"], 5e-3, "d loss / d y_hat")
    torch.testing.assert_close(lat.grad.cpu(), f["glatent"], rtol=1e-3, atol=1e-8)


@pytest.mark.parametrize("reproducible", [False, True])
def test_train_step_captured_in_a_hip_graph_equals_eager(dev, reproducible):
    """the whole train_psp step (pSp forward, pSpLoss, backward, FusedAdam) captured once with torch.cuda.graph and replayed must do
    what the eager loop does: same losses step by step (fixed noise buffers; device-side Adam step count, accumulators zeroed inside
    the graph, deferred loss logs).  bench.py's C5 leg runs this way (5 500 launches per step are CPU-bound at 4 images per GPU).
    reproducible = the library's deterministic mode (no fp32 atomics from several workgroups): the replayed graph and the eager loop then
    agree BIT FOR BIT -- every loss, the step count and the weights after four steps; in the default mode the atomics' arrival order
    differs from run to run and the bounds below are the measured drift."""
    from face_mask_inpaint_amd import functional as FF
    from face_mask_inpaint_amd.modules.psp.criteria import pSpLoss
    from face_mask_inpaint_amd.modules.psp.psp import pSp
    from face_mask_inpaint_amd.optim import FusedAdam

    def run(graph, n_steps=4):
        torch.manual_seed(11)
        opts = types.SimpleNamespace(output_size=64, encoder_type="GradualStyleEncoder", train_decoder=True, use_attention=True, pt_ckpt_path=None,
                                     stylegan_weights=None, learn_in_w=False, start_from_latent_avg=True, decoder_dtype="bf16")
        net = pSp(opts).to(dev).train()
        net.latent_avg = torch.zeros(opts.n_styles, 512, device=dev)
        crit = pSpLoss(types.SimpleNamespace(id_lambda=0, lpips_lambda=0, l2_lambda=1.0, style_lambda=0, lpips_lambda_ref=0, l2_lambda_ref=1.0, cx_lambda=0,
                                             w_norm_lambda=0.005, start_from_latent_avg=True))
        crit.defer_logs = True
        opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, capturable=True)
        g = torch.Generator().manual_seed(5)
        x, ref, y = (torch.rand(2, 3, 256, 256, generator=g).to(dev) * 2 - 1 for _ in range(3))
        m = torch.zeros(2, 256, 256, device=dev)
        m[:, 100:220, 60:200] = 1

        def step():
            y_hat, latent = net(x, ref=ref, src_mask=m, return_latents=True, randomize_noise=False)
            loss, ld, _ = crit(x, y, y_hat, latent, latent_avg=net.latent_avg, ref=ref, mask=m)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        losses = []
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    losses.append(float(step()))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                static_loss = step()
            for _ in range(n_steps - 2):
                cg.replay()
                losses.append(float(static_loss))
        else:
            for _ in range(n_steps):
                losses.append(float(step()))
        step_count = int(opt.param_groups[0]["step_dev"])
        w = net.encoder.styles[0].linear.weight.detach().float().cpu().clone()
        return losses, step_count, w

    with FF.deterministic(reproducible):
        le, se, we = run(False)
        lg, sg, wg = run(True)
    assert se == sg == 4  # the capture itself executes nothing; every replay advances the device-side step count
    if reproducible:
        assert le == lg, (le, lg)
        assert torch.equal(we, wg)
        assert le[-1] != le[0]
        return
    # both runs add split reductions with fp32 atomics in a different order every time, and Adam's first updates (~ lr * sign(g)) turn that rounding
    # noise into different weights: the first two losses agree to 2e-3, later ones drift apart (observed 4e-3 at the third step in one run of three)
    for i, (a, b) in enumerate(zip(le, lg)):
        assert abs(a - b) <= (2e-3 if i < 2 else 1e-2) * abs(a), (le, lg)
    assert le[-1] != le[0]  # the optimiser really moved the weights
    # Adam's update is lr * g / (|g| + eps)-like in the first steps: an element whose gradient is rounding noise moves by up to lr per step in
    # EITHER direction in either run, so two runs can end 2 * steps * lr apart on such an element (observed 6.1e-4; almost all are below 1e-5)
    dw = (we - wg).abs()
    assert float(dw.max()) <= 2 * 4 * 1e-4 * 1.02 and float(dw.median()) <= 2e-5, (float(dw.max()), float(dw.median()))
