"""one bench_psp.train_leg (argv: decoder dtype, size, batch, train_decoder 0/1) for rocprofv3 --kernel-trace --stats: wall time
per step printed, so that the kernel-time sum of the trace shows how much of the step is launch-bound"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench_psp as B
dd, size, batch, td = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
graph = len(sys.argv) > 5 and sys.argv[5] == "graph"
steps = 6
dt, summ = B.train_leg(torch.device("cuda:0"), dd, size, batch, steps, 2, train_decoder=td, graph=graph)
print(f"{dd} {size} bs{batch} train_decoder={td} graph={graph}: {dt / steps * 1e3:.1f} ms per step (wall), {batch * steps / dt:.1f} images/s; profiled step: {summ}")
