"""bisects over GEMM-family launch indices: which launches must use the register-staged kernel for the step-0 G gradients
to agree with the v1 build (argv[1] = candidate lib with FMI_DMA_OFF_RANGE support, argv[2] = v1 reference build)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T
from face_mask_inpaint_amd import functional as FF, _lib
dev = torch.device("cuda:0")
cand, ref = _lib.Library(sys.argv[1]), _lib.Library(sys.argv[2])
counter = ctypes.c_long.in_dll(cand.cdll, "fmi_debug_launch_counter")
fx = torch.load(os.path.join(ROOT, "tests/golden/picnet_train_tiny.pt"), weights_only=True)
s = fx["step0"]


def grads(lib, rng=None):
    _lib._LIB = lib
    if rng is None:
        os.environ.pop("FMI_DMA_OFF_RANGE", None)
    else:
        os.environ["FMI_DMA_OFF_RANGE"] = "%d:%d" % rng
    counter.value = 0
    G, D, gopt, optG, optD = T._tiny_models(fx, dev)
    gopt.lambda_cx = gopt.lambda_perc = gopt.lambda_style = gopt.lambda_g = 0.0
    cap = {}
    orig = optG.step
    def step(closure=None):
        cap.update({n: p.grad.detach().clone() for n, p in G.named_parameters() if p.grad is not None})
        return orig(closure)
    optG.step = step
    m = FF.binarise_mask(s["mask"].to(dev))
    gen = G(s["src"].to(dev), s["ref"].to(dev), src_mask=m, eps=(s["eps_p"].to(dev), s["eps_q"].to(dev)))
    gopt(D, s["src"].to(dev), s["gt"].to(dev), s["ref"].to(dev), gen, m)
    torch.cuda.synchronize()
    return cap, counter.value


want, _ = grads(ref)
def err(rng):
    got, n = grads(cand, rng)
    e = max(float((got[k] - v).abs().max() / (v.abs().max() + 1e-30)) for k, v in want.items() if v.ndim > 1)
    return e, n
e_all, n = err((0, 0))
e_none, _ = err((0, 1 << 30))
print("launches", n, "err all-DMA %.2e, all-v1 %.2e" % (e_all, e_none))
lo, hi = 0, n  # find smallest prefix [0, hi) of v1 launches that fixes it, then the single culprit
while hi - lo > 1:
    mid = (lo + hi) // 2
    e, _ = err((0, mid))
    print("v1 for [0,%d): err %.2e" % (mid, e))
    if e < 2e-3: hi = mid
    else: lo = mid
print("first fixing prefix ends at", hi, "-> culprit launch index", hi - 1)
e, _ = err((hi - 1, hi))
print("only launch %d on v1: err %.2e" % (hi - 1, e))
