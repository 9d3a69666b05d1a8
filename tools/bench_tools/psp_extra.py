"""bench.py's `extra` block alone (C3 / C5 train_psp legs with the bf16 decoder: images/s, ModulatedConv2d bf16 TFLOP/s, in-decoder upfirdn2d / noise+bias+act GB/s)"""
import json, sys
import torch
sys.path.insert(0, "/root/repo")
import bench_psp as B
print(json.dumps(B.extra_block(torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 4)), flush=True)
