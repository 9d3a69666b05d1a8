import gc, sys, collections
import torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
G, D, gopt = bench.build_models(dev, 1)
b = bench.synthetic(8, 256, 1234, dev)
for i in range(12):
    bench.train_step(G, D, gopt, b)
torch.cuda.synchronize(); gc.collect()
ts = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and tuple(o.shape) == (8, 1024)]
print(len(ts), "tensors of shape (8, 1024)")
def describe(o, depth=0, seen=None):
    seen = seen or set()
    if depth > 5 or id(o) in seen:
        return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if r is ts or isinstance(r, type(sys._getframe())):
            continue
        name = type(r).__name__
        extra = ""
        if isinstance(r, dict):
            extra = str([k for k, v in r.items() if v is o][:3])
        elif isinstance(r, (list, tuple)):
            extra = f"len {len(r)}"
        print("  " * depth + f"<- {name} {extra}")
        if name in ("dict", "list", "tuple", "cell") or "Backward" in name or "ctx" in name.lower():
            describe(r, depth + 1, seen)
describe(ts[0])
