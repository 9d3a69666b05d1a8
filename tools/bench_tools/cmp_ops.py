"""functional ops (forward + backward) under two library builds: argv[1] candidate, argv[2] reference"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
libs = [_lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])]
g = torch.Generator().manual_seed(0)


def run(fn, *shapes):
    xs = [torch.randn(*s, generator=g) for s in shapes]
    res = []
    for lib in libs:
        _lib._LIB = lib
        ins = [x.to(dev).requires_grad_(True) for x in xs]
        out = fn(*ins)
        outs = out if isinstance(out, (tuple, list)) else [out]
        tot = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=dev).view(o.shape)).sum() for o in outs)
        tot.backward()
        torch.cuda.synchronize()
        res.append([o.detach() for o in outs] + [i.grad for i in ins])
    errs = [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(*res) if a is not None]
    return errs


bad = 0
for c in (8, 16, 32, 64):
    for p in (16, 64, 256, 1024, 4096):
        e = run(lambda x: FF.gram_matrix(x), (2, p, c))
        e2 = run(lambda x, y: FF.contextual_loss(x, y.detach()), (2, min(p, 1024), c), (2, min(p, 1024), c))
        flag = max(e + e2) > 1e-4
        bad += flag
        print("c", c, "p", p, "gram", ["%.1e" % v for v in e], "cx", ["%.1e" % v for v in e2], "<-- BAD" if flag else "")
for (n, t, d, cv) in [(2, 64, 4, 16), (2, 256, 8, 32), (2, 1024, 16, 16), (2, 4096, 4, 8)]:
    e = run(lambda q, v1, v2: FF.self_attention(q, [v1, v2]), (n, t, d), (n, t, cv), (n, t, cv))
    flag = max(e) > 1e-4
    bad += flag
    print("attn", (n, t, d, cv), ["%.1e" % v for v in e], "<-- BAD" if flag else "")
print("BAD" if bad else "OK", bad)
