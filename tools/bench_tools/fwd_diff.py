"""first module whose forward output differs between the two builds (argv[1] candidate, argv[2] reference), tiny config"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
libs = [_lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])]
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]
outs = []
grs = []
for lib in libs:
    _lib._LIB = lib
    G, D, gopt, optG, optD = T._tiny_models(fx, dev)
    rec = []
    live = []
    gr = {}
    for fname in ("conv2d", "conv_transpose2d", "self_attention", "instance_norm_act", "add", "leaky_relu", "avg_pool", "vae_sample", "guide_blend_cat", "resize_bilinear"):
        def mk(fname=fname, orig=getattr(FF, fname) if not hasattr(FF, "_orig_" + fname) else getattr(FF, "_orig_" + fname)):
            setattr(FF, "_orig_" + fname, orig)
            def w(*a, **k):
                out = orig(*a, **k)
                o = out[0] if isinstance(out, (tuple, list)) else out
                idx = len(rec)
                rec.append((fname, tuple(a[0].shape) if torch.is_tensor(a[0]) else None, o.detach().clone()))
                live.append(o.detach())
                if o.requires_grad:
                    o.register_hook(lambda g, idx=idx: gr.__setitem__(idx, g.detach().clone()))
                return out
            return w
        setattr(FF, fname, mk())
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    FF._orig_l1 = getattr(FF, "_orig_l1", FF.l1_loss)
    loss = FF._orig_l1(FF.to_nhwc(gen), FF.to_nhwc(s["gt"].to(dev)))
    loss.backward()
    torch.cuda.synchronize()
    for i, ((fn_, sh_, snap), lv) in enumerate(zip(rec, live)):
        if not torch.equal(snap, lv):
            nbad = int((snap != lv).sum())
            print("MODIFIED AFTER FORWARD: op %d %s in %s: %d of %d elements changed, max |delta| %.3e" % (i, fn_, sh_, nbad, snap.numel(), float((snap - lv).abs().max())))
    outs.append(rec)
    grs.append(gr)
print(len(outs[0]), len(outs[1]))
for i, ((na, ta, a), (nb, tb, b)) in enumerate(zip(*outs)):
    e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    if e > 1e-6 or i < 12:
        print("%3d %.2e %s in %s out %s" % (i, e, na, ta, tuple(a.shape)))

print("backward (cotangent of each op output), last op first:")
for i in sorted(grs[0], reverse=True):
    a, b = grs[0][i], grs[1][i]
    e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    if e > 1e-5 or i >= 108:
        print("%3d %.2e d/d out of %s in %s  |g|max %.3e  nonzero frac %.3f" % (i, e, outs[0][i][0], outs[0][i][1], float(b.abs().max()), float((b != 0).float().mean())))
print("sign statistics of leaky_relu / instance_norm_act outputs (build A vs B):")
for i, ((na, ta, a), (nb, tb, b)) in enumerate(zip(*outs)):
    if na in ("leaky_relu", "instance_norm_act") and i > 60:
        mx = float(b.abs().max())
        frac_small = float((b.abs() < 1e-5 * mx).float().mean())
        flips = float(((a > 0) != (b > 0)).float().mean())
        print("%3d %s out %s  max %.3e  median|y| %.3e  frac(|y|<1e-5 max) %.3e  sign flips %.3e" % (i, na, tuple(b.shape), mx, float(b.abs().median()), frac_small, flips))
a, b = grs[0][115], grs[1][115]
d = (a - b).abs()
idx = (d > 1e-3 * b.abs().max()).nonzero()
print("op115 cotangent: differing elements", idx.shape[0], "of", a.numel())
print("first positions", idx[:12].tolist())
r = (a / b)[d > 1e-3 * b.abs().max()]
print("ratio a/b of differing: min %.3f max %.3f median %.3f" % (float(r.min()), float(r.max()), float(r.median())))
g116a, g116b = grs[0][116], grs[1][116]
xa, xb = outs[0][115][2], outs[1][115][2]
ii = tuple(idx[0].tolist())
print("at", ii, "g116", float(g116a[ii]), float(g116b[ii]), "x115", float(xa[ii]), float(xb[ii]), "out", float(a[ii]), float(b[ii]))
import collections
print("rows (y) histogram of differing:", collections.Counter(idx[:, 1].tolist()).most_common(8))
print("cols (x) histogram of differing:", collections.Counter(idx[:, 2].tolist()).most_common(8))
