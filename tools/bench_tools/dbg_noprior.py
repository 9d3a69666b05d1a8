import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_infer as T
from face_mask_inpaint_amd.modules.model import ReferenceFill
from face_mask_inpaint_amd import functional as FF
from oracle import picnet_cpu as O
dev = torch.device("cuda:0")
f = torch.load(os.path.join(ROOT, "tests/golden/picnet_infer.pt"), weights_only=True)["variants"]
G = T._load(ReferenceFill(None, dict(T.ENC), dict(T.DEC), use_att=True, out_size=(100, 90)), f["sd0"], dev)
P = O.prepare_params(f["sd0"])
with torch.no_grad():
    o_raw = G(f["src"].to(dev), f["ref"].to(dev), src_mask=f["mask"].to(dev), no_prior=True, resize=False)
    w_raw = O.reference_fill_forward(P, f["src"], f["ref"], f["mask"], None, None, no_prior=True, resize=False, **dict(enc_layers=5, enc_L=2, enc_z_nc=8, dec_layers=5, dec_L=0))
    print("no_prior raw err", float((o_raw.cpu() - w_raw).abs().max()))
    a = FF.to_nchw(FF.resize_bilinear(FF.to_nhwc(w_raw.to(dev)), 218, 178)).cpu()
    b = O.scale_img(w_raw, (218, 178))
    print("resize err on identical input", float((a - b).abs().max()), "vs fixture", float((b - f["no_prior"]).abs().max()))
    x = torch.rand(2, 3, 256, 256)
    a = FF.to_nchw(FF.resize_bilinear(FF.to_nhwc(x.to(dev)), 218, 178)).cpu()
    print("resize rand err", float((a - O.scale_img(x, (218, 178))).abs().max()))
    a = FF.to_nchw(FF.resize_bilinear(FF.to_nhwc(x.to(dev)), 32, 32)).cpu()
    print("resize 256->32 err", float((a - O.scale_img(x, (32, 32))).abs().max()))
    a = FF.to_nchw(FF.resize_bilinear(FF.to_nhwc(x[:, :, :64, :64].contiguous().to(dev)), 218, 178)).cpu()
    print("resize 64->218x178 err", float((a - O.scale_img(x[:, :, :64, :64], (218, 178))).abs().max()))
