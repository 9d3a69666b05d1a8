"""the generator's last block at its real size (bs 8): ResBlockDecoder(64 -> 32) 512^2 -> 1024^2 = conv 64 -> 32, ConvTranspose 32 -> 32 with the
bypass ConvTranspose 64 -> 32 as residual, forward + backward, per-call-site table (functional.PROFILE) -- the 32 / 64-channel layers that are
12 ms of the C2 step.  Under rocprofv3 --kernel-trace --stats it shows which kernels they are."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF

dev = torch.device("cuda:0")
PAIR = not os.environ.get("NO_PAIR")  # NO_PAIR=1: main path and bypass as two ConvTranspose2d calls
torch.manual_seed(0)
n, s = 8, 512
x = torch.randn(n, s, s, 64, device=dev, requires_grad=True)
w1 = (torch.randn(32, 64, 3, 3, device=dev) * 0.05).requires_grad_(True)
w2 = (torch.randn(32, 32, 3, 3, device=dev) * 0.05).requires_grad_(True)   # ConvTranspose2d weight [in][out][kh][kw]
wb = (torch.randn(64, 32, 3, 3, device=dev) * 0.05).requires_grad_(True)
b1, b2, bb = (torch.zeros(32, device=dev, requires_grad=True) for _ in range(3))
gy = torch.randn(n, 2 * s, 2 * s, 32, device=dev)


def step():
    for t in (x, w1, w2, wb, b1, b2, bb):
        t.grad = None
    pw1, pw2, pwb = FF.prepare_weights([(w1, None, None), (w2, None, None, True), (wb, None, None, True)])
    h = FF.conv2d(x, pw1, b1, pad=1, in_act=("apply", 0.1))
    ha = FF.leaky_relu(h, 0.1)
    if PAIR and FF.conv_transpose2d_pair_ok(ha, pw2, x, pwb):
        y = FF.conv_transpose2d_pair(ha, pw2, x, pwb, FF.add(b2, bb))
    else:
        sc = FF.conv_transpose2d(x, pwb, bb)
        y = FF.conv_transpose2d(ha, pw2, b2, residual=sc)
    y.backward(gy)


for _ in range(2):
    step()
torch.cuda.synchronize()
FF.PROFILE = []
step()
torch.cuda.synchronize()
recs, FF.PROFILE = FF.PROFILE, None
tot = 0.0
for tag, fl, s_, e_ in recs:
    ms = s_.elapsed_time(e_)
    tot += ms
    print("%8.3f ms %7.1f TFLOP/s  %s" % (ms, fl / ms / 1e9, tag))
print("sum %.3f ms" % tot)
