"""memory growth per training step: census of live CUDA tensors by (shape, dtype) between two steps"""
import gc, sys, collections
import torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
G, D, gopt = bench.build_models(dev, 1)
b = bench.synthetic(8, 256, 1234, dev)
def census():
    c = collections.Counter()
    st = set()
    for o in gc.get_objects():
        try:
            if isinstance(o, torch.Tensor) and o.is_cuda:
                c[(tuple(o.shape), str(o.dtype))] += 1
                st.add(o.untyped_storage().data_ptr())
        except Exception:
            pass
    return c, len(st)
out = []
for i in range(26):
    bench.train_step(G, D, gopt, b)
    if i in (10, 25):
        torch.cuda.synchronize()
        gc.collect()
        c, ns = census()
        out.append(c)
        print(i, "alloc GB", round(torch.cuda.memory_allocated() / 2**30, 3), "tensors", sum(c.values()), "storages", ns)
diff = {k: out[1][k] - out[0].get(k, 0) for k in out[1] if out[1][k] != out[0].get(k, 0)}
for k, v in sorted(diff.items(), key=lambda kv: -abs(kv[1]))[:20]:
    print(v, k)
