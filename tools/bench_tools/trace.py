import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FMI_DMA_OFF_RANGE"] = "0:0"; os.environ["FMI_DMA_TRACE"] = "1"
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
G, D, gopt, optG, optD = T._tiny_models(fx, dev)
from face_mask_inpaint_amd import _lib
L = _lib.lib()
orig = L.conv2d_fwd_f32
cnt = [0]
def traced(d, x, wf, bias, res, y, act, bw, bws, st):
    dd = d._obj
    print("conv_fwd call", cnt[0], {f[0]: getattr(dd, f[0]) for f in dd._fields_}, "bias", bool(bias and bias.value), "res", bool(res and res.value), "act", act,
          "x%16", (x.value or 0) % 16, "y%16", (y.value or 0) % 16, file=sys.stderr)
    cnt[0] += 1
    return orig(d, x, wf, bias, res, y, act, bw, bws, st)
L.conv2d_fwd_f32 = traced
m = FF.binarise_mask(s["mask"].to(dev))
gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
torch.cuda.synchronize()
