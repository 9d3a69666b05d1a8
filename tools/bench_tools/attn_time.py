"""fused self-attention forward + backward at the decoder's shape (T 16384, d 64, C 256, batch 8): ms and TFLOP/s; the PMC runs of
profiles/round1_pmc_attention.json profile this script"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
n, t, d, c = 8, 16384, 64, 256
q = (torch.randn(n, t, d, device=dev) * 0.3).requires_grad_(True)
v = torch.randn(n, t, c, device=dev, requires_grad=True)
g = torch.randn(n, t, c, device=dev)
def run():
    (o,) = FF.self_attention(q, [v])
    o.backward(g)
    q.grad = None
    v.grad = None
for _ in range(2):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
for _ in range(4):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 4
fl = 2.0 * n * t * t * (d + c) + 2.0 * n * t * t * (3 * d + 2 * c)
with torch.no_grad():
    FF.self_attention(q, [v])
    torch.cuda.synchronize()
    s.record()
    for _ in range(4):
        FF.self_attention(q, [v])
    e.record()
    torch.cuda.synchronize()
fwd = s.elapsed_time(e) / 4
print(f"fwd {fwd:.2f} ms  bwd {ms - fwd:.2f} ms  fwd+bwd {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
