"""GPU micro-benchmarks of the MFMA GEMM family at the hot-path shapes (TFLOP/s per call site)."""
import ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
from face_mask_inpaint_amd import functional as FF, _lib
lib = _lib.lib(); dev = torch.device('cuda:0')

def timeit(fn, flops, name, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:46s} {ms:8.3f} ms  {flops/ms/1e9:7.1f} TFLOP/s", flush=True)

def conv_case(n, h, c, k, ks=3, stride=1, pad=1):
    x = torch.randn(n, h, h, c, device=dev); w = torch.randn(ks*ks, c, k, device=dev) * 0.05
    wt = w.permute(0, 2, 1).contiguous()
    d, oh, ow = FF.conv_desc(n, h, h, c, k, ks, ks, stride, pad)
    y = torch.empty(n, oh, ow, k, device=dev); gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    fl = 2.0 * n * oh * ow * k * c * ks * ks
    st = FF._st()
    timeit(lambda: lib.conv2d_fwd_f32(C.byref(d), FF._p(x), FF._p(w), None, None, FF._p(y), 0, 1, 0, st), fl, f"conv_fwd   n{n} {h}x{h} {c}->{k} k{ks}s{stride}")
    timeit(lambda: lib.conv2d_dgrad_f32(C.byref(d), FF._p(gy), FF._p(wt), None, None, FF._p(dx), 1, 0, st), fl, f"conv_dgrad n{n} {h}x{h} {c}->{k} k{ks}s{stride}")
    timeit(lambda: lib.conv2d_wgrad_f32(C.byref(d), FF._p(x), FF._p(gy), FF._p(dw), None, 1, 0, st), fl, f"conv_wgrad n{n} {h}x{h} {c}->{k} k{ks}s{stride}")

def gemm_case(m, n, k, b, ta, tb, name):
    a = torch.randn(b, k, m, device=dev) if ta else torch.randn(b, m, k, device=dev)
    bb = torch.randn(b, n, k, device=dev) if tb else torch.randn(b, k, n, device=dev)
    c = torch.zeros(b, m, n, device=dev)
    sa = (1, m) if ta else (k, 1); sb = (1, k) if tb else (n, 1)
    timeit(lambda: FF.gemm_raw(FF._p(a), FF._p(bb), FF._p(c), m, n, k, sa, sb, (n, 1), b, (m*k, k*n, m*n)), 2.0*m*n*k*b, name)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "conv"):
        conv_case(8, 128, 256, 256); conv_case(8, 256, 128, 128); conv_case(8, 512, 64, 64); conv_case(8, 1024, 32, 32)
        conv_case(8, 224, 64, 64); conv_case(8, 56, 256, 256); conv_case(8, 256, 128, 256, 3, 2, 1); conv_case(8, 32, 128, 128)
    if which in ("all", "gemm"):
        gemm_case(4096, 4096, 4096, 1, 0, 0, "gemm 4096^3 (K,X)")
        gemm_case(4096, 4096, 4096, 1, 0, 1, "gemm 4096^3 (K,K)")
        gemm_case(4096, 4096, 4096, 1, 1, 0, "gemm 4096^3 (X,X)")
        gemm_case(128, 16384, 64, 8, 0, 1, "attn_qk  128x16384x64 b8")
        gemm_case(128, 256, 16384, 8, 0, 0, "attn_pv  128x256x16384 b8 (split-K)")
        gemm_case(16384, 256, 128, 8, 1, 0, "attn_dv  16384x256x128 b8")
        gemm_case(128, 16384, 256, 8, 0, 1, "attn_dp  128x16384x256 b8")
        gemm_case(128, 64, 16384, 8, 0, 0, "attn_dq  128x64x16384 b8 (split-K)")
        gemm_case(16384, 64, 128, 8, 1, 0, "attn_dk  16384x64x128 b8")
