// Pooling and bilinear resampling on NHWC fp32 tensors.  Bandwidth kernels: one thread per group of four
// channels (float4) of an output pixel when C % 4 == 0, otherwise one thread per element.
#include "common.h"

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int V>
struct VecT;
template <>
struct VecT<1> {
  typedef float T;
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ void add(float& a, float b) { a += b; }
  static __device__ __forceinline__ void scale(float& a, float s) { a *= s; }
};
template <>
struct VecT<4> {
  typedef float4 T;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void add(float4& a, float4 b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  static __device__ __forceinline__ void scale(float4& a, float s) {
    a.x *= s; a.y *= s; a.z *= s; a.w *= s;
  }
};

// ---- k x k average pooling -------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                      int CV, int k, int64_t total) {
  typedef typename VecT<V>::T T;
  const int OH = H / k, OW = W / k;
  const float inv = 1.f / (float)(k * k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CV);
    int64_t r = i / CV;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    T s = VecT<V>::zero();
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b)
        VecT<V>::add(s, reinterpret_cast<const T*>(x)[(((int64_t)n * H + oy * k + a) * W + ox * k + b) * CV + c]);
    VecT<V>::scale(s, inv);
    reinterpret_cast<T*>(y)[i] = s;
  }
}
template <int V>
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H,
                                                          int W, int CV, int k, int64_t total) {
  typedef typename VecT<V>::T T;
  const int OH = H / k, OW = W / k;
  const float inv = 1.f / (float)(k * k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CV);
    int64_t r = i / CV;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    T v = VecT<V>::zero();
    if (yy / k < OH && xx / k < OW) {
      v = reinterpret_cast<const T*>(gy)[(((int64_t)n * OH + yy / k) * OW + xx / k) * CV + c];
      VecT<V>::scale(v, inv);
    }
    reinterpret_cast<T*>(gx)[i] = v;
  }
}

// Global average pool (SEModule, helpers.py:56-72: AdaptiveAvgPool2d(1)): one output pixel per image.  The windowed kernel above
// would give each thread a k*k-long strided walk; here 256 threads stream a slice of the image as float4 (consecutive threads =
// consecutive channel chunks), combine through LDS and meet the other slices of the image through fp32 atomics (y zeroed first).
__global__ void __launch_bounds__(256) global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t HW, int C4,
                                                            int64_t rows_per_block, float inv) {
  __shared__ float4 part[256];
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > HW) r1 = HW;
  const float4* xp = reinterpret_cast<const float4*>(x) + (int64_t)n * HW * C4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {
    const float4 v0 = xp[r * C4 + cg], v1 = xp[(r + RL) * C4 + cg];
    s.x += v0.x + v1.x; s.y += v0.y + v1.y; s.z += v0.z + v1.z; s.w += v0.w + v1.w;
  }
  for (; r < r1; r += RL) {
    const float4 v = xp[r * C4 + cg];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C4) {
    float4 t = part[threadIdx.x];
    for (int l = 1; l < RL; ++l) {
      const float4 v = part[l * C4 + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float* o = y + ((int64_t)n * C4 + threadIdx.x) * 4;
    atomicAdd(o + 0, t.x * inv);
    atomicAdd(o + 1, t.y * inv);
    atomicAdd(o + 2, t.z * inv);
    atomicAdd(o + 3, t.w * inv);
  }
}

extern "C" int fmi_avgpool_f32(const float* x, float* y, int N, int H, int W, int C, int k, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || H / k <= 0 || W / k <= 0) return FMI_ERR_BAD_ARG;
  {
    const int C4 = C / 4;
    if (k == H && k == W && k >= 8 && C % 4 == 0 && C4 <= 256 && (C4 & (C4 - 1)) == 0 && al16(x) && al16(y) && N <= 65535) {
      const int64_t HW = (int64_t)H * W;
      int64_t blocks = ceil_div64(HW, 128);
      if (blocks * N > 2048) blocks = ceil_div64(2048, N);
      if (fmi_det()) blocks = 1;  // reproducible mode: one block per image
      const int64_t rpb = ceil_div64(HW, blocks);
      blocks = ceil_div64(HW, rpb);
      if (hipMemsetAsync(y, 0, (size_t)N * C * sizeof(float), (hipStream_t)stream) != hipSuccess) return FMI_ERR_LAUNCH;
      hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, y, HW, C4, rpb,
                         1.f / (float)HW);
      return fmi_launch_status();
    }
  }
  const bool v4 = (C % 4 == 0) && al16(x) && al16(y);
  const int CV = v4 ? C / 4 : C;
  const int64_t total = (int64_t)N * (H / k) * (W / k) * CV;
  if (v4) hipLaunchKernelGGL((avgpool_kernel<4>), dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, CV, k, total);
  else hipLaunchKernelGGL((avgpool_kernel<1>), dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, CV, k, total);
  return fmi_launch_status();
}
extern "C" int fmi_avgpool_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int k, void* stream) {
  if (!gy || !gx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || H / k <= 0 || W / k <= 0) return FMI_ERR_BAD_ARG;
  const bool v4 = (C % 4 == 0) && al16(gx) && al16(gy);
  const int CV = v4 ? C / 4 : C;
  const int64_t total = (int64_t)N * H * W * CV;
  if (v4) hipLaunchKernelGGL((avgpool_bwd_kernel<4>), dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, CV, k, total);
  else hipLaunchKernelGGL((avgpool_bwd_kernel<1>), dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, CV, k, total);
  return fmi_launch_status();
}

// ---- 2 x 2 max pooling (floor mode); the gradient goes to the FIRST maximum in row-major window order ----
__global__ void __launch_bounds__(256) maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                       int C, int64_t total) {
  const int OH = H / 2, OW = W / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const float* p = x + (((int64_t)n * H + oy * 2) * W + ox * 2) * C + c;
    float m = p[0];
    m = fmaxf(m, p[C]);
    m = fmaxf(m, p[(int64_t)W * C]);
    m = fmaxf(m, p[(int64_t)W * C + C]);
    y[i] = m;
  }
}
__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ gx, int H, int W, int C, int64_t total) {
  const int OH = H / 2, OW = W / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    const int oy = yy >> 1, ox = xx >> 1;
    float g = 0.f;
    if (oy < OH && ox < OW) {
      const float* p = x + (((int64_t)n * H + oy * 2) * W + ox * 2) * C + c;
      const float v0 = p[0], v1 = p[C], v2 = p[(int64_t)W * C], v3 = p[(int64_t)W * C + C];
      int arg = 0;
      float m = v0;
      if (v1 > m) { m = v1; arg = 1; }
      if (v2 > m) { m = v2; arg = 2; }
      if (v3 > m) { m = v3; arg = 3; }
      if (arg == ((yy & 1) * 2 + (xx & 1))) g = gy[(((int64_t)n * OH + oy) * OW + ox) * C + c];
    }
    gx[i] = g;
  }
}
extern "C" int fmi_maxpool2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, total);
  return fmi_launch_status();
}
extern "C" int fmi_maxpool2_bwd_f32(const float* x, const float* gy, float* gx, int N, int H, int W, int C, void* stream) {
  if (!x || !gy || !gx || N <= 0 || H < 2 || W < 2 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, H, W, C, total);
  return fmi_launch_status();
}

// ---- bilinear, align_corners=True (same source-index arithmetic as ATen's upsample_bilinear2d) ----
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_of(int o, int in, int out) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float r = scale * (float)o;
  Lerp l;
  l.i0 = (int)r;
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.l1 = r - (float)l.i0;
  l.l0 = 1.f - l.l1;
  return l;
}
__global__ void __launch_bounds__(256) resize_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                     int OH, int OW, const float* __restrict__ mean,
                                                     const float* __restrict__ stdv, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const Lerp ly = lerp_of(oy, H, OH), lx = lerp_of(ox, W, OW);
    const float* b = x + (int64_t)n * H * W * C + c;
    const float v00 = b[((int64_t)ly.i0 * W + lx.i0) * C], v01 = b[((int64_t)ly.i0 * W + lx.i1) * C];
    const float v10 = b[((int64_t)ly.i1 * W + lx.i0) * C], v11 = b[((int64_t)ly.i1 * W + lx.i1) * C];
    float v = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    if (mean) v = (v - mean[c]) / stdv[c];
    y[i] = v;
  }
}
__global__ void __launch_bounds__(256) resize_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H, int W,
                                                         int C, int OH, int OW, const float* __restrict__ stdv,
                                                         int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const Lerp ly = lerp_of(oy, H, OH), lx = lerp_of(ox, W, OW);
    float g = gy[i];
    if (stdv) g = g / stdv[c];
    float* b = gx + (int64_t)n * H * W * C + c;
    atomicAdd(b + ((int64_t)ly.i0 * W + lx.i0) * C, ly.l0 * lx.l0 * g);
    atomicAdd(b + ((int64_t)ly.i0 * W + lx.i1) * C, ly.l0 * lx.l1 * g);
    atomicAdd(b + ((int64_t)ly.i1 * W + lx.i0) * C, ly.l1 * lx.l0 * g);
    atomicAdd(b + ((int64_t)ly.i1 * W + lx.i1) * C, ly.l1 * lx.l1 * g);
  }
}
// reproducible form of the adjoint: every INPUT pixel gathers from the output pixels whose interpolation footprint contains it, in a
// fixed order (gx is written, not accumulated).  Candidates: outputs around i / scale, tested with the forward's own lerp_of.
__device__ __forceinline__ void resize_cands(int i, int in, int out, int& lo, int& hi) {
  const float inv = in > 1 ? (float)(out - 1) / (float)(in - 1) : 0.f;
  lo = (int)((float)(i - 1) * inv) - 2;
  hi = (int)((float)(i + 1) * inv) + 2;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__global__ void __launch_bounds__(256) resize_bwd_gather_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H, int W, int C,
                                                                int OH, int OW, const float* __restrict__ stdv, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    int ylo, yhi, xlo, xhi;
    resize_cands(iy, H, OH, ylo, yhi);
    resize_cands(ix, W, OW, xlo, xhi);
    const float* b = gy + (int64_t)n * OH * OW * C + c;
    float s = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const Lerp ly = lerp_of(oy, H, OH);
      const float wy = (ly.i0 == iy ? ly.l0 : 0.f) + (ly.i1 == iy ? ly.l1 : 0.f);
      if (ly.i0 != iy && ly.i1 != iy) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const Lerp lx = lerp_of(ox, W, OW);
        if (lx.i0 != ix && lx.i1 != ix) continue;
        const float wx = (lx.i0 == ix ? lx.l0 : 0.f) + (lx.i1 == ix ? lx.l1 : 0.f);
        float g = b[((int64_t)oy * OW + ox) * C];
        if (stdv) g = g / stdv[c];
        // the scatter form adds l_y * l_x * g per corner; a pixel that is BOTH corners of an axis (the clamped last row) gets both weights
        if (ly.i0 == iy && ly.i1 == iy) {
          if (lx.i0 == ix) s += ly.l0 * lx.l0 * g + ly.l1 * lx.l0 * g;
          if (lx.i1 == ix) s += ly.l0 * lx.l1 * g + ly.l1 * lx.l1 * g;
        } else {
          const float w1 = ly.i0 == iy ? ly.l0 : ly.l1;
          if (lx.i0 == ix) s += w1 * lx.l0 * g;
          if (lx.i1 == ix) s += w1 * lx.l1 * g;
        }
        (void)wx;
        (void)wy;
      }
    }
    gx[i] = s;
  }
}
extern "C" int fmi_resize_bilinear_f32(const float* x, float* y, int N, int H, int W, int C, int OH, int OW,
                                       const float* ch_mean, const float* ch_std, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return FMI_ERR_BAD_ARG;
  if ((ch_mean == nullptr) != (ch_std == nullptr)) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * OH * OW * C;
  hipLaunchKernelGGL(resize_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, OH, OW,
                     ch_mean, ch_std, total);
  return fmi_launch_status();
}
extern "C" int fmi_resize_bilinear_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int OH, int OW,
                                           const float* ch_std, void* stream) {
  if (!gy || !gx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return FMI_ERR_BAD_ARG;
  if (fmi_det()) {  // reproducible mode: gather form (gx overwritten)
    const int64_t tin = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(resize_bwd_gather_kernel, dim3(fmi_bw_grid(tin, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, C, OH, OW, ch_std, tin);
    return fmi_launch_status();
  }
  const int64_t total = (int64_t)N * OH * OW * C;
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, C, OH,
                     OW, ch_std, total);
  return fmi_launch_status();
}

// ---- nn.AdaptiveAvgPool2d for ANY input / output size (model.py:79 on a 218 x 178 decoder output, id_loss.py:19: 188 -> 112) ----
// window of output i along an axis of length L -> OL: [floor(i L / OL), ceil((i + 1) L / OL)); windows overlap when L % OL != 0 and
// have length 1 (replication) when OL > L.  One thread per output element (vectorised over channels when C % 4 == 0).
__device__ __forceinline__ int aap_lo(int i, int L, int OL) { return (int)(((int64_t)i * L) / OL); }
__device__ __forceinline__ int aap_hi(int i, int L, int OL) { return (int)(((int64_t)(i + 1) * L + OL - 1) / OL); }
template <int V>
__global__ void __launch_bounds__(256) adaptive_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH,
                                                               int OW, int CV, int64_t total) {
  typedef typename VecT<V>::T T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CV);
    int64_t r = i / CV;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const int y0 = aap_lo(oy, H, OH), y1 = aap_hi(oy, H, OH), x0 = aap_lo(ox, W, OW), x1 = aap_hi(ox, W, OW);
    T s = VecT<V>::zero();
    for (int a = y0; a < y1; ++a)
      for (int b = x0; b < x1; ++b) VecT<V>::add(s, reinterpret_cast<const T*>(x)[(((int64_t)n * H + a) * W + b) * CV + c]);
    VecT<V>::scale(s, 1.f / (float)((y1 - y0) * (x1 - x0)));
    reinterpret_cast<T*>(y)[i] = s;
  }
}
// gradient: an input pixel gathers from every output window that contains it (at most a handful; candidates around h OL / L)
template <int V>
__global__ void __launch_bounds__(256) adaptive_avgpool_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int H, int W,
                                                                   int OH, int OW, int CV, int64_t total) {
  typedef typename VecT<V>::T T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % CV);
    int64_t r = i / CV;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    // outputs whose window contains yy: lo(o) <= yy < hi(o); lo is non-decreasing in o, so scan from the first candidate
    int oy0 = (int)(((int64_t)yy * OH) / H);
    while (oy0 > 0 && aap_hi(oy0 - 1, H, OH) > yy) --oy0;
    int ox0 = (int)(((int64_t)xx * OW) / W);
    while (ox0 > 0 && aap_hi(ox0 - 1, W, OW) > xx) --ox0;
    T s = VecT<V>::zero();
    for (int oy = oy0; oy < OH && aap_lo(oy, H, OH) <= yy; ++oy) {
      if (aap_hi(oy, H, OH) <= yy) continue;
      const int hy = aap_hi(oy, H, OH) - aap_lo(oy, H, OH);
      for (int ox = ox0; ox < OW && aap_lo(ox, W, OW) <= xx; ++ox) {
        if (aap_hi(ox, W, OW) <= xx) continue;
        const int hx = aap_hi(ox, W, OW) - aap_lo(ox, W, OW);
        T v = reinterpret_cast<const T*>(gy)[(((int64_t)n * OH + oy) * OW + ox) * CV + c];
        VecT<V>::scale(v, 1.f / (float)(hy * hx));
        VecT<V>::add(s, v);
      }
    }
    reinterpret_cast<T*>(gx)[i] = s;
  }
}
extern "C" int fmi_adaptive_avgpool_f32(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return FMI_ERR_BAD_ARG;
  const bool v4 = C % 4 == 0 && al16(x) && al16(y);
  const int64_t total = (int64_t)N * OH * OW * (v4 ? C / 4 : C);
  if (v4)
    hipLaunchKernelGGL(adaptive_avgpool_kernel<4>, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, C / 4, total);
  else
    hipLaunchKernelGGL(adaptive_avgpool_kernel<1>, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, C, total);
  return fmi_launch_status();
}
extern "C" int fmi_adaptive_avgpool_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int OH, int OW, void* stream) {
  if (!gy || !gx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return FMI_ERR_BAD_ARG;
  const bool v4 = C % 4 == 0 && al16(gy) && al16(gx);
  const int64_t total = (int64_t)N * H * W * (v4 ? C / 4 : C);
  if (v4)
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<4>, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, OH, OW, C / 4, total);
  else
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<1>, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, H, W, OH, OW, C, total);
  return fmi_launch_status();
}

// ---- nn.MaxPool2d(k, stride) without padding, floor mode (LPIPS' AlexNet trunk: k 3 stride 2, criteria/lpips/networks.py) ----
// forward also records the flat window position of the FIRST maximum (torch's tie rule) so that the backward is a gather-free scatter
__global__ void __launch_bounds__(256) maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ arg,
                                                      int H, int W, int C, int OH, int OW, int k, int stride, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    float best = -INFINITY;
    int bi = 0;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const float v = x[(((int64_t)n * H + oy * stride + a) * W + ox * stride + b) * C + c];
        if (v > best || v != v) {  // ATen's rule: a later NaN replaces the running maximum
          best = v;
          bi = a * k + b;
        }
      }
    y[i] = best;
    if (arg) arg[i] = bi;
  }
}
// windows overlap when stride < k, so the scatter uses atomics (gx zeroed by the caller)
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ gy, const int32_t* __restrict__ arg,
                                                          float* __restrict__ gx, int H, int W, int C, int OH, int OW, int k, int stride,
                                                          int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const int bi = arg[i];
    atomicAdd(gx + (((int64_t)n * H + oy * stride + bi / k) * W + ox * stride + bi % k) * C + c, gy[i]);
  }
}
// reproducible form: every input pixel gathers from the windows that selected it, in a fixed order (gx written, not accumulated)
__global__ void __launch_bounds__(256) maxpool_bwd_gather_kernel(const float* __restrict__ gy, const int32_t* __restrict__ arg,
                                                                 float* __restrict__ gx, int H, int W, int C, int OH, int OW, int k, int stride,
                                                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    int oy0 = (y - k + stride) / stride, ox0 = (x - k + stride) / stride;  // ceil((y - k + 1) / stride) for y - k + 1 > 0
    if (y - k + 1 <= 0) oy0 = 0;
    if (x - k + 1 <= 0) ox0 = 0;
    int oy1 = y / stride, ox1 = x / stride;
    if (oy1 > OH - 1) oy1 = OH - 1;
    if (ox1 > OW - 1) ox1 = OW - 1;
    float s = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        const int64_t o = (((int64_t)n * OH + oy) * OW + ox) * C + c;
        if (arg[o] == (y - oy * stride) * k + (x - ox * stride)) s += gy[o];
      }
    gx[i] = s;
  }
}
extern "C" int fmi_maxpool_f32(const float* x, float* y, int32_t* argmax, int N, int H, int W, int C, int k, int stride, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || k <= 0 || stride <= 0 || H < k || W < k) return FMI_ERR_BAD_ARG;
  const int OH = (H - k) / stride + 1, OW = (W - k) / stride + 1;
  const int64_t total = (int64_t)N * OH * OW * C;
  hipLaunchKernelGGL(maxpool_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, H, W, C, OH, OW, k, stride, total);
  return fmi_launch_status();
}
extern "C" int fmi_maxpool_bwd_f32(const float* gy, const int32_t* argmax, float* gx, int N, int H, int W, int C, int k, int stride,
                                   void* stream) {
  if (!gy || !argmax || !gx || N <= 0 || C <= 0 || k <= 0 || stride <= 0 || H < k || W < k) return FMI_ERR_BAD_ARG;
  const int OH = (H - k) / stride + 1, OW = (W - k) / stride + 1;
  const int64_t total = (int64_t)N * OH * OW * C;
  if (fmi_det()) {  // reproducible mode: gather form
    const int64_t tin = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_gather_kernel, dim3(fmi_bw_grid(tin, 256)), dim3(256), 0, (hipStream_t)stream, gy, argmax, gx, H, W, C, OH, OW, k,
                       stride, tin);
    return fmi_launch_status();
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, argmax, gx, H, W, C, OH, OW, k, stride, total);
  return fmi_launch_status();
}

// ---- argmax over the channel axis of an NHWC map, written as a float mask (PICNet_inference.py:100-101:
// mask_detector(src, 'train').argmax(1).float()); index work: the FIRST maximum wins like torch.argmax, bit exact ----
__global__ void __launch_bounds__(256) argmax_channels_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int64_t pixels) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    const float* row = x + p * C;
    float best = row[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const float v = row[c];
      if (v > best || (v != v && best == best)) {  // NaN counts as the maximum (torch.argmax)
        best = v;
        bi = c;
      }
    }
    out[p] = (float)bi;
  }
}
extern "C" int fmi_argmax_channels_f32(const float* x, float* out, int64_t pixels, int C, void* stream) {
  if (!x || !out || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_channels_kernel, dim3(fmi_bw_grid(pixels, 256)), dim3(256), 0, (hipStream_t)stream, x, out, C, pixels);
  return fmi_launch_status();
}
