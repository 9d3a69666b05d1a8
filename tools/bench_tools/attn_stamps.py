"""phase timeline of the fused attention backward (attn_bwd2_x6_kernel<64, 8>): needs the diagnostic build
`FILES=attention tools/bench_tools/build_flags.sh attstamp -DFMI_ATT_STAMP`; prints the s_memtime stamps of query tile 100 for wave 0 of
workgroup (0, 0), relative to the tile's start ("where a query tile of the attention backward spends its cycles")"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FMI_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libfmi_attstamp.so"))
from face_mask_inpaint_amd import functional as FF
dev = torch.device("cuda:0")
n, t, d, c = 8, 16384, 64, 256
torch.manual_seed(0)
q = (torch.randn(n, t, d, device=dev) * 0.2).requires_grad_(True)
v = torch.randn(n, t, c, device=dev).requires_grad_(True)
for _ in range(3):
    q.grad = v.grad = None
    (o,) = FF.self_attention(q, [v])
    o.square().mean().backward()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib = C.CDLL(os.environ["FMI_LIB_PATH"])
lib.fmi_debug_attn_stamps.argtypes = [C.c_void_p]
assert lib.fmi_debug_attn_stamps(buf) == 0
names = ["top", "S done", "dP done", "P / dS done", "dK half 1 done", "dK half 2 done, next tile's loads issued", "dQ MFMAs done", "partials stored", "after barrier", "partials read",
         "next tile staged", "sums, atomics issued + barrier"]
print(" ".join("%s=%d" % (names[i], buf[i] - buf[0]) for i in range(12)))
