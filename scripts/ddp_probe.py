"""2-rank probe of DataParallelOptimizer on ONE GPU (gloo): torch-only toy model, prints progress per rank."""
import os, sys, faulthandler
faulthandler.enable()
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_mask_inpaint_amd import distributed as fd

rank = int(os.environ["RANK"])
def log(*a):
    print(f"[r{rank}]", *a, file=sys.stderr, flush=True)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
log("init ok")
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 8)).cuda()
fd.broadcast_parameters([m])
log("broadcast ok")
opt = fd.DataParallelOptimizer(torch.optim.Adam(m.parameters(), lr=1e-3), bucket_bytes=4096)
x = torch.randn(16, 64, device="cuda")
for i in range(3):
    opt.zero_grad()
    m(x).square().mean().backward()
    log("backward ok", i, "inflight", len(opt._inflight), "open", len(opt._open))
    opt.launch()
    log("launch ok", i)
    opt.step()
    log("step ok", i)
torch.cuda.synchronize()
dist.barrier()
log("done")
dist.destroy_process_group()
