import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from oracle import picnet_cpu as O
from face_mask_inpaint_amd.modules.loss import VGGLoss
from face_mask_inpaint_amd.modules.model import ReferenceFill
from face_mask_inpaint_amd.modules.pluralistic_model import network
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip(), flush=True)
except Exception as e:
    print("no cgroup cpu.max", e, flush=True)
os.system("grep -m1 'model name' /proc/cpuinfo")
for th in (int(sys.argv[1]) if len(sys.argv) > 1 else 16,):
    torch.set_num_threads(th)
    torch.manual_seed(0)
    G = ReferenceFill(None, dict(bench.ENC), dict(bench.DEC), use_att=True, out_size=(256, 256))
    D = network.define_d(**bench.DISC)
    PG, PD = O.prepare_params(G.state_dict()), O.prepare_params(D.state_dict())
    PV = O.prepare_params(VGGLoss().state_dict(), frozen=True)
    og = torch.optim.Adam(O.unique_trainable(PG), lr=1e-5); od = torch.optim.Adam(O.unique_trainable(PD), lr=1e-5)
    src, ref, gt, mask, eps_p, eps_q = O.synthetic_batch(1, 256, seed=1234, feat_hw=32, z_nc=128)
    for it in range(2):
        t0 = time.time()
        with torch.no_grad():
            pass
        O.train_step(PG, PD, PV, og, od, src, gt, ref, mask, eps_p, eps_q, out_size=(256, 256))
        print(f"threads {th} step {it}: {time.time()-t0:.2f} s", flush=True)
