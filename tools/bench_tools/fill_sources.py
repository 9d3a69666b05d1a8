"""who launches the fill / add kernels of a PICNet training step: torch profiler with python stacks, grouped by the innermost
face_mask_inpaint_amd frame (or 'autograd engine' when the op has no python caller)"""
import sys, collections
import torch
sys.path.insert(0, "/root/repo")
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
G, D, gopt = bench.build_models(dev, 1)
batch = bench.synthetic(8, 256, 1234, dev)
for _ in range(2):
    bench.train_step(G, D, gopt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    bench.train_step(G, D, gopt, batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat"):
        where = "autograd engine / no python frame"
        for fr in ev.stack:
            if "face_mask_inpaint_amd" in fr or "bench.py" in fr:
                where = fr.split("face_mask_inpaint_amd/")[-1][:90]
                break
        cnt[(ev.name, str(ev.input_shapes)[:80] + " | " + where[:40])] += 1
for (name, where), c in cnt.most_common(40):
    print(f"{c:5d}  {name:12s} {where}")
