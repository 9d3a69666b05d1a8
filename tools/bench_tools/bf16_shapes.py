"""per-launch table of the bf16 decoder forward + backward (event-timed)"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import functional as FF
from face_mask_inpaint_amd.modules.psp.stylegan2.model import Generator
dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gen = Generator(size, 512, 8, compute_dtype=torch.bfloat16).to(dev)
lat = torch.randn(n, gen.n_latent, 512, device=dev, requires_grad=True)
def run():
    img, _ = gen([lat], input_is_latent=True, randomize_noise=True)
    img.square().mean().backward()
run(); run()
torch.cuda.synchronize()
FF.PROFILE = []
run()
torch.cuda.synchronize()
recs, FF.PROFILE = FF.PROFILE, None
rows = [(s.elapsed_time(e), t, f) for t, f, s, e in recs if "_bf16|" in t]
tot = sum(r[0] for r in rows)
print(f"total {tot:.3f} ms, {sum(r[2] for r in rows) / tot / 1e9:.0f} TF")
for ms, t, f in sorted(rows, reverse=True)[:45]:
    print(f"{ms:8.3f} ms {f / ms / 1e9:7.0f} TF  {t}")
