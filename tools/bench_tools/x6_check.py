"""accuracy of the dense GEMM / convolution of each library variant against float64, next to time_core.py's rates"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_mask_inpaint_amd import functional as FF, _lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
a = torch.randn(2048, 4096, device=dev) * torch.exp(torch.randn(2048, 1, device=dev) * 3)
b = torch.randn(4096, 1024, device=dev) * torch.exp(torch.randn(4096, 1, device=dev))
ref = a.double() @ b.double()
den = (a.double().abs() @ b.double().abs())
tor = (a @ b).double()
print("torch fp32 matmul: max |err| / (|a||b|) = %.3e, mean %.3e" % (float(((tor - ref).abs() / den).max()), float(((tor - ref).abs() / den).mean())))
n, h, ci, co, k = 4, 64, 256, 256, 3
x = torch.randn(n, h, h, ci, device=dev); w = torch.randn(k * k, ci, co, device=dev) * 0.05
d, oh, ow = FF.conv_desc(n, h, h, ci, co, k, k, 1, k // 2)
yref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.view(k, k, ci, co).permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
yden = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().abs(), w.view(k, k, ci, co).permute(3, 2, 0, 1).double().abs(), padding=1).permute(0, 2, 3, 1)
for path in sys.argv[1:]:
    lib = _lib.Library(path)
    c = torch.empty(2048, 1024, device=dev)
    lib.gemm_f32(FF._p(a), FF._p(b), FF._p(c), 2048, 1024, 4096, 4096, 1, 1024, 1, 1024, 1, 1, 0, 0, 0, 1.0, 0.0, None, st)
    e = (c.double() - ref).abs() / den
    y = torch.empty(n, oh, ow, co, device=dev)
    lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y), 0, 1, 0, st)
    ey = (y.double() - yref).abs() / yden
    print("%-24s gemm: max %.3e mean %.3e signed-mean %.2e | conv3x3: max %.3e mean %.3e" % (
        os.path.basename(path), float(e.max()), float(e.mean()), float(((c.double() - ref) / den).mean()), float(ey.max()), float(ey.mean())))
