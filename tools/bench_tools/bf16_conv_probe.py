"""bf16 convolution kernels against torch on bf16-rounded operands (fp32 accumulation); prints errors and timings"""
import ctypes as C, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from face_mask_inpaint_amd import _lib, functional as FF

lib = _lib.lib()
dev = torch.device("cuda:0")


def run(n, c, k, h, w, ks, stride, pad, colscale=False):
    g = torch.Generator().manual_seed(c + k + h)
    x = torch.randn(n, c, h, w, generator=g).bfloat16()
    wt_ = (torch.randn(k, c, ks, ks, generator=g) / (c * ks * ks) ** 0.5).bfloat16()
    y_ref = F.conv2d(x.float(), wt_.float(), stride=stride, padding=pad)
    gy = torch.randn(y_ref.shape, generator=g).bfloat16()
    dx_ref = torch.autograd.grad(F.conv2d(x.float().requires_grad_(True), wt_.float(), stride=stride, padding=pad), [], None) if False else None
    xr = x.float().requires_grad_(True)
    F.conv2d(xr, wt_.float(), stride=stride, padding=pad).backward(gy.float())
    dx_ref = xr.grad
    wr = wt_.float().requires_grad_(True)
    F.conv2d(x.float(), wr, stride=stride, padding=pad).backward(gy.float())
    dw_ref = wr.grad.permute(2, 3, 1, 0).reshape(ks * ks, c, k)
    d, oh, ow = FF.conv_desc(n, h, w, c, k, ks, ks, stride, pad, 0)
    wf = wt_.float().permute(2, 3, 1, 0).reshape(ks * ks, c, k).contiguous().to(dev)   # [tap][C][K]
    wtp = wt_.float().permute(2, 3, 0, 1).reshape(ks * ks, k, c).contiguous().to(dev)  # [tap][K][C]
    wnk = torch.empty(k, ks * ks, c, dtype=torch.bfloat16, device=dev)
    wck = torch.empty(c, ks * ks, k, dtype=torch.bfloat16, device=dev)
    st = FF._st()
    lib.pack_weight_bf16(FF._p(wf), FF._p(wnk), ks * ks, c, k, st)
    lib.pack_weight_bf16(FF._p(wtp), FF._p(wck), ks * ks, k, c, st)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    gh = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    cs = (torch.rand(n, k, generator=g) + 0.5).to(dev) if colscale else None
    y = torch.full((n, oh, ow, k), float("nan"), dtype=torch.bfloat16, device=dev)
    lib.conv2d_fwd_bf16(C.byref(d), FF._p(xh), FF._p(wnk), FF._p(cs), FF._p(y), None, 0, st)
    yr = y_ref.permute(0, 2, 3, 1)
    if colscale:
        yr = yr * cs.cpu().view(n, 1, 1, k)
    e1 = (y.float().cpu() - yr).abs().max().item() / yr.abs().max().item()
    dx = torch.full((n, h, w, c), float("nan"), dtype=torch.bfloat16, device=dev)
    lib.conv2d_dgrad_bf16(C.byref(d), FF._p(gh), FF._p(wck), None, FF._p(dx), None, 0, st)
    dr = dx_ref.permute(0, 2, 3, 1)
    e2 = (dx.float().cpu() - dr).abs().max().item() / dr.abs().max().item()
    dwf = torch.zeros(ks * ks, c, k, device=dev)
    lib.conv2d_wgrad_bf16(C.byref(d), FF._p(xh), FF._p(gh), FF._p(dwf), st)
    e3 = (dwf.cpu() - dw_ref).abs().max().item() / dw_ref.abs().max().item()
    torch.cuda.synchronize()
    t = []
    for fn in (lambda: lib.conv2d_fwd_bf16(C.byref(d), FF._p(xh), FF._p(wnk), None, FF._p(y), None, 0, st),
               lambda: lib.conv2d_dgrad_bf16(C.byref(d), FF._p(gh), FF._p(wck), None, FF._p(dx), None, 0, st),
               lambda: lib.conv2d_wgrad_bf16(C.byref(d), FF._p(xh), FF._p(gh), FF._p(dwf), st)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20)
    fl = 2.0 * n * oh * ow * k * c * ks * ks
    print(f"n{n} {c}->{k} {h}x{w} k{ks}s{stride}p{pad}: fwd err {e1:.2e} dgrad err {e2:.2e} wgrad err {e3:.2e} | fwd {t[0]*1e3:.3f} ms {fl/t[0]/1e12:.0f} TF  dgrad {t[1]*1e3:.3f} ms {fl/t[1]/1e12:.0f} TF  wgrad {t[2]*1e3:.3f} ms {fl/t[2]/1e12:.0f} TF", flush=True)


if __name__ == "__main__":
    run(2, 64, 64, 12, 10, 3, 1, 1, True)
    run(2, 32, 96, 9, 11, 3, 1, 1)
    run(1, 128, 32, 16, 16, 1, 1, 0)
    run(2, 64, 128, 8, 8, 3, 2, 0)     # adjoint of a stride-2 conv = ConvTranspose phases
    run(2, 96, 64, 13, 9, 3, 2, 1)
    run(16, 512, 512, 64, 64, 3, 1, 1)
    run(16, 256, 256, 128, 128, 3, 1, 1)
    run(16, 128, 128, 256, 256, 3, 1, 1)
    run(16, 512, 512, 16, 16, 3, 1, 1)
    run(16, 512, 512, 16, 16, 3, 1, 1)
    run(4, 32, 32, 512, 512, 3, 1, 1)
