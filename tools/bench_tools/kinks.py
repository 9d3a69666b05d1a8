"""LeakyReLU inputs of the decoder of the tiny golden config (step argv[1]): elements whose SIGN differs between HIP, the oracle in
fp32 and the oracle in fp64 (a flipped kink changes one derivative from 1 to 0.1)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF
from face_mask_inpaint_amd.weights import weight_scope
from face_mask_inpaint_amd.modules.pluralistic_model.base_function import _conv
from face_mask_inpaint_amd.modules.pluralistic_model.external_function import run_conv
from oracle import picnet_cpu as O
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 0
f2 = dict(fx); f2["G_sd0"], f2["D_sd0"] = fx[f"G_sd{step}"], fx[f"D_sd{step}"]
s = fx[f"step{step}"]; cfg = fx["config"]
G, D, gopt, optG, optD = T._tiny_models(f2, dev)
pre = {}
def oracle(dt, tag):
    c = lambda t: t.to(dt)
    P = O.prepare_params(f2["G_sd0"], dtype=dt)
    with torch.no_grad():
        mask = c(O.binarise_mask(s["mask"]))
        sd, sf = O.res_encoder(P, "src_encoder", c(s["src"]), "src", 5, cfg["enc_L"], cfg["enc_z_nc"])
        rd, rf = O.res_encoder(P, "ref_encoder", c(s["ref"]), "ref", 5, cfg["enc_L"], cfg["enc_z_nc"])
        m = O.scale_img(mask.unsqueeze(1), sf.shape[-2:])
        enc = O.example_guided_attention(P, "attention", m, sf, rf)
        z = O.get_z(sd, rd, c(s["eps_p"]), c(s["eps_q"]))
        out = enc + O.res_block(P, "decoder.generator", z)
        for i in range(5):
            pfx = f"decoder.decoder{i}"
            a0 = O.inst_norm(P, pfx + ".model.0", out)
            h = O.sn_conv(P, pfx + ".conv1", O.lrelu(a0), padding=1)
            a1 = O.inst_norm(P, pfx + ".model.3", h)
            out = O.sn_conv_transpose(P, pfx + ".conv2", O.lrelu(a1)) + O.sn_conv_transpose(P, pfx + ".bypass", out)
            pre[(tag, f"dec{i}.in0")], pre[(tag, f"dec{i}.in1")] = a0.double(), a1.double()
            if i == 1:
                out = O.auto_attn(P, "decoder.attn1", out)
        pre[(tag, "out4.in")] = out.double()
oracle(torch.float64, "o64"); oracle(torch.float32, "o32")
with torch.no_grad(), weight_scope(G):
    src, ref = FF.to_nhwc(s["src"].to(dev)), FF.to_nhwc(s["ref"].to(dev))
    o_src, src_feat = G.src_encoder.nhwc_raw(src)
    o_ref, ref_feat = G.ref_encoder.nhwc_raw(ref)
    n, fh, fw, _ = src_feat.shape
    md = FF.resize_bilinear(FF.binarise_mask(s["mask"].to(dev)).unsqueeze(-1), fh, fw).view(n, fh, fw)
    enc_g = G.attention.nhwc(md, src_feat, ref_feat)
    zg = FF.vae_sample(o_src, o_ref, FF.to_nhwc(s["eps_q"].to(dev)), FF.to_nhwc(s["eps_p"].to(dev)))
    out = FF.add(enc_g, G.decoder.generator.nhwc(zg))
    cv = lambda t: t.cpu().permute(0, 3, 1, 2).double()
    for i in range(5):
        blk = getattr(G.decoder, f"decoder{i}")
        a0 = FF.instance_norm_act(out, blk.model[0].weight, blk.model[0].bias, 1e-5, 1.0)
        h = run_conv(_conv(blk.conv1), FF.leaky_relu(a0, 0.1))
        a1 = FF.instance_norm_act(h, blk.model[3].weight, blk.model[3].bias, 1e-5, 1.0)
        pre[("hip", f"dec{i}.in0")], pre[("hip", f"dec{i}.in1")] = cv(a0), cv(a1)
        out = blk.nhwc(out)
        if i == 1:
            out = G.decoder.attn1.nhwc(out)
    pre[("hip", "out4.in")] = cv(out)
names = sorted({k[1] for k in pre})
for nm in names:
    a64, a32, ah = pre[("o64", nm)], pre[("o32", nm)], pre[("hip", nm)]
    f32 = (torch.sign(a32) != torch.sign(a64)); fh_ = (torch.sign(ah) != torch.sign(a64))
    print("%-9s n %7d  max|err| hip %.1e o32 %.1e | sign flips vs fp64: hip %d o32 %d | exact zeros: hip %d o32 %d o64 %d | min|x64| %.1e  |x64| at hip flips %s" % (
        nm, a64.numel(), float((ah - a64).abs().max()), float((a32 - a64).abs().max()), int(fh_.sum()), int(f32.sum()),
        int((ah == 0).sum()), int((a32 == 0).sum()), int((a64 == 0).sum()), float(a64.abs().min()), ["%.1e" % v for v in a64[fh_].abs().tolist()[:6]]))
