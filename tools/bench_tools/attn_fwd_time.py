"""fused self-attention FORWARD at the decoder's shape (T 16384, d 64, C 256, batch 8): accuracy against float64 on a slice of queries, ms, TFLOP/s"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
n, t, d, c = 8, 16384, 64, 256
q = torch.randn(n, t, d, device=dev) * 0.3
v = torch.randn(n, t, c, device=dev)
with torch.no_grad():
    (o,) = FF.self_attention(q, [v])
    qs = q[0, :512].double()
    ref = torch.softmax(qs @ q[0].double().T, -1) @ v[0].double()
    print("max |err| vs float64 on 512 queries: %.3e (max |o| %.3e)" % (float((o[0, :512].double() - ref).abs().max()), float(ref.abs().max())))
    for _ in range(2):
        FF.self_attention(q, [v])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(5):
        FF.self_attention(q, [v])
    e.record()
    torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
fl = 2.0 * n * t * t * (d + c)
print(f"fwd {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
