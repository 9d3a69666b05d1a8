"""bf16 StyleGAN2 256^2 decoder forward + backward, bs 16 (for rocprofv3 --kernel-trace --stats)"""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd.modules.psp.stylegan2.model import Generator
dev = torch.device("cuda:0")
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
torch.manual_seed(0)
gen = Generator(256, 512, 8, compute_dtype=dt).to(dev)
lat = torch.randn(16, gen.n_latent, 512, device=dev, requires_grad=True)
def run():
    img, _ = gen([lat], input_is_latent=True, randomize_noise=True)
    img.square().mean().backward()
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print(f"{dt}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per forward+backward (bs 16)")
