// InstanceNorm2d(affine=True) on NHWC fp32, fused with the LeakyReLU that always follows it on the
// hot path (base_function.py:348-356).  Bandwidth kernels.  Statistics are accumulated per thread in
// fp32 over a short pixel run, then in fp64 across threads / workgroups (fp64 atomics), so the
// E[x^2]-E[x]^2 form does not lose precision at 512x512 planes.
//
// Layout of a workgroup: 256 threads = PL pixel lanes x CG channel groups (4 channels each), CG = C/4.
#include "common.h"

// Element types: float, or bf16 held as uint16_t (the mixed-precision IR-SE50 body of the pSp encoder: bf16 activations, fp32
// statistics / parameters / arithmetic).  Four channels per thread either way: a 16-byte or an 8-byte access.
__device__ __forceinline__ float nb_bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t nb_f2bf(float f) {  // round to nearest even
  const uint32_t u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float4 nld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 nld4(const uint16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(nb_bf2f(v.x & 0xffffu), nb_bf2f(v.x >> 16), nb_bf2f(v.y & 0xffffu), nb_bf2f(v.y >> 16));
}
__device__ __forceinline__ void nst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void nst4(uint16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(nb_f2bf(v.x) | (nb_f2bf(v.y) << 16), nb_f2bf(v.z) | (nb_f2bf(v.w) << 16));
}

// pixels per workgroup chunk: at least 64, grown until at most `cap` workgroups exist.  With fp64 atomics (no workspace) every
// workgroup adds 2 C values onto the same 2 C addresses, and those serialise: more workgroups were SLOWER there (cap 2048 from 512-row
// chunks).  With a partials workspace each workgroup stores its own row and a second launch adds the rows, so the chunk can shrink
// until ~1024 workgroups stream (the BatchNorm planes of the pSp encoder -- 8192 .. 32768 rows of 256 - 512 channels -- ran on 16 - 64
// workgroups whose two pixel lanes walked 256 rows each: 50 us of load latency per launch, 200 launches per step).
static int rows_per_block(int N, int HW, int64_t cap, int64_t r0) {
  int64_t r = r0;
  while ((int64_t)N * ((HW + r - 1) / r) > cap && r < (1 << 20)) r *= 2;
  return (int)r;
}
// out[g][i] = sum_{p < nparts} ws[(g * nparts + p) * width + i].  64 columns x 16 row lanes per workgroup, 8 independent loads in
// flight per thread (one thread per column walking 1024 rows took 37 us: pure load latency, 200 launches per pSp step)
__global__ void __launch_bounds__(1024) sum_parts_f64_kernel(const double* __restrict__ ws, double* __restrict__ out, int nparts, int width) {
  __shared__ double part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const double* p = ws + (int64_t)blockIdx.y * nparts * width + i;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < width) {
    int q = ty;
    for (; q + 7 * 16 < nparts; q += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += p[(int64_t)(q + 16 * u) * width];
    }
    for (; q < nparts; q += 16) a[0] += p[(int64_t)q * width];
  }
  part[ty][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (ty == 0 && i < width) {
    double t = 0;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += part[l][tx];
    out[(int64_t)blockIdx.y * width + i] = t;
  }
}

// sums[n][c][0..1] += (sum f0, sum f1) where (f0,f1) = fn(x, g) per element
template <int MODE, typename T>  // 0: (x, x*x)   1: backward reductions (g', g'*xhat)
__global__ void __launch_bounds__(256) in_reduce_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, double* __restrict__ sums, int HW,
                                                        int C, float slope, int rpb, int to_parts) {
  __shared__ float red[256 * 8];
  const int CG = C >> 2, PL = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * rpb;
  int p1 = p0 + rpb;
  if (p1 > HW) p1 = HW;
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  float mean[4] = {0, 0, 0, 0}, rstd[4] = {1, 1, 1, 1}, gm[4] = {1, 1, 1, 1}, bt[4] = {0, 0, 0, 0};
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = cg * 4 + e;
      mean[e] = stats[((int64_t)n * C + c) * 2];
      rstd[e] = stats[((int64_t)n * C + c) * 2 + 1];
      gm[e] = gamma[c];
      bt[e] = beta[c];
    }
  }
  if (pl < PL) {
    // four pixels per iteration: independent 16-byte loads in flight (a single dependent load per thread reached 1.4 TB/s)
    auto body = [&](const float4 v, const float4 g4) {
      const float xv[4] = {v.x, v.y, v.z, v.w};
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0[e] += xv[e];
          a1[e] += xv[e] * xv[e];
        }
      } else {
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mean[e]) * rstd[e];
          const float pre = xh * gm[e] + bt[e];
          const float gp = pre > 0.f ? gv[e] : gv[e] * slope;
          a0[e] += gp;
          a1[e] += gp * xh;
        }
      }
    };
    int p = p0 + pl;
    for (; p + 3 * PL < p1; p += 4 * PL) {
      float4 v[4], g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t o = ((int64_t)n * HW + p + u * PL) * C + cg * 4;
        v[u] = nld4(x + o);
        g4[u] = MODE == 1 ? nld4(gy + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) body(v[u], g4[u]);
    }
    for (; p < p1; p += PL) {
      const int64_t o = ((int64_t)n * HW + p) * C + cg * 4;
      body(nld4(x + o), MODE == 1 ? nld4(gy + o) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[threadIdx.x * 8 + e] = a0[e];
    red[threadIdx.x * 8 + 4 + e] = a1[e];
  }
  __syncthreads();
  if (threadIdx.x < CG) {  // thread cg sums over the pixel lanes in fp64
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int l = 0; l < PL; ++l) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] += red[(l * CG + threadIdx.x) * 8 + e];
        s1[e] += red[(l * CG + threadIdx.x) * 8 + 4 + e];
      }
    }
    if (to_parts) {  // row (n, block) of the partials workspace
      double* d = sums + (((int64_t)n * gridDim.x + blockIdx.x) * C + threadIdx.x * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[2 * e] = s0[e], d[2 * e + 1] = s1[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double* d = sums + ((int64_t)n * C + threadIdx.x * 4 + e) * 2;
        atomicAdd(d, s0[e]);
        atomicAdd(d + 1, s1[e]);
      }
    }
  }
}
// launches the reduction; with a workspace of ws_doubles >= N * C * 2 doubles the sums are WRITTEN (partials + one adding launch),
// without one they are accumulated onto the caller-zeroed buffer by fp64 atomics
template <int MODE, typename T>
static void launch_in_reduce(const T* x, const T* gy, const float* stats, const float* gamma, const float* beta, double* sums,
                             int N, int HW, int C, float slope, double* ws, int64_t ws_doubles, hipStream_t st) {
  if (ws && ws_doubles >= (int64_t)N * C * 2 && (((uintptr_t)ws) & 15) == 0) {
    int64_t cap = ws_doubles / ((int64_t)C * 2);
    if (cap > 1024) cap = 1024;
    const int rpb = rows_per_block(N, HW, cap, 64);
    const int blocks = (HW + rpb - 1) / rpb;
    hipLaunchKernelGGL((in_reduce_kernel<MODE, T>), dim3(blocks, N), dim3(256), 0, st, x, gy, stats, gamma, beta, ws, HW, C, slope, rpb, 1);
    hipLaunchKernelGGL(sum_parts_f64_kernel, dim3((C * 2 + 63) / 64, N), dim3(1024), 0, st, (const double*)ws, sums, blocks, C * 2);
    return;
  }
  const int rpb = fmi_det() ? HW : rows_per_block(N, HW, 2048, 512);  // reproducible mode: one block per sample, one contribution per address
  hipLaunchKernelGGL((in_reduce_kernel<MODE, T>), dim3((HW + rpb - 1) / rpb, N), dim3(256), 0, st, x, gy, stats, gamma, beta, sums, HW, C, slope,
                     rpb, 0);
}

__global__ void __launch_bounds__(256) in_finalize_kernel(const double* __restrict__ sums, float* __restrict__ stats, int NC,
                                                          int HW, float eps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NC) return;
  const double m = sums[2 * i] / HW;
  double var = sums[2 * i + 1] / HW - m * m;
  if (var < 0) var = 0;
  stats[2 * i] = (float)m;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

static int check_c(int C) {
  const int cg = C / 4;
  if (C % 4 != 0 || cg < 1 || cg > 256) return FMI_ERR_UNSUPPORTED;  // one thread per 4-channel group; 256 % CG lanes idle
  return FMI_OK;
}

template <typename T>
static int stats_impl(const T* x, double* sums, float* stats, int N, int HW, int C, float eps, double* ws, int64_t ws_doubles, void* stream) {
  if (!x || !sums || !stats || N <= 0 || HW <= 0 || C <= 0 || ((uintptr_t)x & (4 * sizeof(T) - 1))) return FMI_ERR_BAD_ARG;
  if (check_c(C)) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  launch_in_reduce<0, T>(x, (const T*)nullptr, nullptr, nullptr, nullptr, sums, N, HW, C, 1.f, ws, ws_doubles, st);
  hipLaunchKernelGGL(in_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const double*)sums, stats, N * C, HW, eps);
  return fmi_launch_status();
}
extern "C" int fmi_instnorm_stats_f32(const float* x, double* sums, float* stats, int N, int HW, int C, float eps, double* ws,
                                      int64_t ws_doubles, void* stream) {
  return stats_impl(x, sums, stats, N, HW, C, eps, ws, ws_doubles, stream);
}
extern "C" int fmi_instnorm_stats_bf16(const uint16_t* x, double* sums, float* stats, int N, int HW, int C, float eps, double* ws,
                                       int64_t ws_doubles, void* stream) {
  return stats_impl(x, sums, stats, N, HW, C, eps, ws, ws_doubles, stream);
}

// y = act((x - mean) * rstd * gamma + beta).  A thread keeps ONE 4-channel group (its six per-channel values live in registers for the
// whole launch) and walks pixels of one sample, four independent 16-byte loads in flight -- the flat grid-stride form re-read
// stats / gamma / beta for every element (24 extra loads per 16 bytes moved) and ran at 2.4-2.8 TB/s.  Same arithmetic per element.
template <typename T>
__global__ void __launch_bounds__(256) in_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, int HW, int C, float slope, int rpb) {
  const int CG = C >> 2, PL = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= PL) return;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * rpb;
  int p1 = p0 + rpb;
  if (p1 > HW) p1 = HW;
  float mean[4], rstd[4], gm[4], bt[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cg * 4 + e;
    mean[e] = stats[((int64_t)n * C + c) * 2];
    rstd[e] = stats[((int64_t)n * C + c) * 2 + 1];
    gm[e] = gamma[c];
    bt[e] = beta[c];
  }
  auto one = [&](int64_t o, const float4 v) {
    const float xv[4] = {v.x, v.y, v.z, v.w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pre = (xv[e] - mean[e]) * rstd[e] * gm[e] + bt[e];
      r[e] = pre > 0.f ? pre : pre * slope;
    }
    nst4(y + o, make_float4(r[0], r[1], r[2], r[3]));
  };
  int p = p0 + pl;
  for (; p + 3 * PL < p1; p += 4 * PL) {
    float4 v[4];
    int64_t o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o[u] = ((int64_t)n * HW + p + u * PL) * C + cg * 4;
      v[u] = nld4(x + o[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(o[u], v[u]);
  }
  for (; p < p1; p += PL) {
    const int64_t o = ((int64_t)n * HW + p) * C + cg * 4;
    one(o, nld4(x + o));
  }
}
template <typename T>
static int apply_impl(const T* x, const float* stats, const float* gamma, const float* beta, T* y, int N, int HW, int C, float slope, void* stream) {
  const uintptr_t al = 4 * sizeof(T) - 1;
  if (!x || !stats || !gamma || !beta || !y || N <= 0 || HW <= 0 || ((uintptr_t)x & al) || ((uintptr_t)y & al)) return FMI_ERR_BAD_ARG;
  if (check_c(C)) return FMI_ERR_UNSUPPORTED;
  const int rpb = rows_per_block(N, HW, 16384, 16);
  hipLaunchKernelGGL((in_apply_kernel<T>), dim3((HW + rpb - 1) / rpb, N), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, y, HW, C, slope,
                     rpb);
  return fmi_launch_status();
}
extern "C" int fmi_instnorm_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, float* y,
                                      int N, int HW, int C, float slope, void* stream) {
  return apply_impl(x, stats, gamma, beta, y, N, HW, C, slope, stream);
}
extern "C" int fmi_instnorm_apply_bf16(const uint16_t* x, const float* stats, const float* gamma, const float* beta, uint16_t* y,
                                       int N, int HW, int C, float slope, void* stream) {
  return apply_impl(x, stats, gamma, beta, y, N, HW, C, slope, stream);
}

template <typename T>
static int bwd_reduce_impl(const T* x, const T* gy, const float* stats, const float* gamma, const float* beta, double* red, int N, int HW, int C,
                           float slope, double* ws, int64_t ws_doubles, void* stream) {
  const uintptr_t al = 4 * sizeof(T) - 1;
  if (!x || !gy || !stats || !gamma || !beta || !red || N <= 0 || HW <= 0 || ((uintptr_t)x & al) || ((uintptr_t)gy & al)) return FMI_ERR_BAD_ARG;
  if (check_c(C)) return FMI_ERR_UNSUPPORTED;
  launch_in_reduce<1, T>(x, gy, stats, gamma, beta, red, N, HW, C, slope, ws, ws_doubles, (hipStream_t)stream);
  return fmi_launch_status();
}
extern "C" int fmi_instnorm_bwd_reduce_f32(const float* x, const float* gy, const float* stats, const float* gamma,
                                           const float* beta, double* red, int N, int HW, int C, float slope, double* ws,
                                           int64_t ws_doubles, void* stream) {
  return bwd_reduce_impl(x, gy, stats, gamma, beta, red, N, HW, C, slope, ws, ws_doubles, stream);
}
extern "C" int fmi_instnorm_bwd_reduce_bf16(const uint16_t* x, const uint16_t* gy, const float* stats, const float* gamma,
                                            const float* beta, double* red, int N, int HW, int C, float slope, double* ws,
                                            int64_t ws_doubles, void* stream) {
  return bwd_reduce_impl(x, gy, stats, gamma, beta, red, N, HW, C, slope, ws, ws_doubles, stream);
}

// gx = rstd*gamma*(g' - mean(g') - xhat*mean(g'*xhat))
template <typename T>
__global__ void __launch_bounds__(256) in_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const double* __restrict__ red,
                                                           T* __restrict__ gx, int HW, int C, float slope, const T* __restrict__ gadd, int rpb) {
  const int CG = C >> 2, PL = 256 / CG;  // the row structure of in_apply_kernel: eight per-channel values per thread, loaded once
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= PL) return;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * rpb;
  int p1 = p0 + rpb;
  if (p1 > HW) p1 = HW;
  const float inv_hw = 1.f / (float)HW;
  float mean[4], rstd[4], gm[4], bt[4], m1[4], m2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cg * 4 + e;
    const int64_t sc = (int64_t)n * C + c;
    mean[e] = stats[sc * 2], rstd[e] = stats[sc * 2 + 1];
    gm[e] = gamma[c], bt[e] = beta[c];
    m1[e] = (float)red[sc * 2] * inv_hw, m2[e] = (float)red[sc * 2 + 1] * inv_hw;
  }
  auto one = [&](int64_t o, const float4 v, const float4 g4, const float4 a) {
    const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - mean[e]) * rstd[e];
      const float pre = xh * gm[e] + bt[e];
      const float gp = pre > 0.f ? gv[e] : gv[e] * slope;
      r[e] = rstd[e] * gm[e] * (gp - m1[e] - xh * m2[e]);
    }
    if (gadd) r[0] += a.x, r[1] += a.y, r[2] += a.z, r[3] += a.w;  // the second consumer's gradient (identity shortcut of an IR block)
    nst4(gx + o, make_float4(r[0], r[1], r[2], r[3]));
  };
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = p0 + pl;
  for (; p + 3 * PL < p1; p += 4 * PL) {
    float4 v[4], g4[4], a[4];
    int64_t o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o[u] = ((int64_t)n * HW + p + u * PL) * C + cg * 4;
      v[u] = nld4(x + o[u]);
      g4[u] = nld4(gy + o[u]);
      a[u] = gadd ? nld4(gadd + o[u]) : z;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(o[u], v[u], g4[u], a[u]);
  }
  for (; p < p1; p += PL) {
    const int64_t o = ((int64_t)n * HW + p) * C + cg * 4;
    one(o, nld4(x + o), nld4(gy + o), gadd ? nld4(gadd + o) : z);
  }
}
__global__ void __launch_bounds__(256) in_param_grad_kernel(const double* __restrict__ red, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int N, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int n = 0; n < N; ++n) {
    s1 += red[((int64_t)n * C + c) * 2];
    s2 += red[((int64_t)n * C + c) * 2 + 1];
  }
  dbeta[c] += (float)s1;
  dgamma[c] += (float)s2;
}
template <typename T>
static int in_bwd_apply_impl(const T* x, const T* gy, const float* stats, const float* gamma, const float* beta, const double* red,
                             T* gx, float* dgamma, float* dbeta, int N, int HW, int C, float slope, const T* gadd, void* stream) {
  const uintptr_t al = 4 * sizeof(T) - 1;
  if (!x || !gy || !stats || !gamma || !beta || !red || !gx || N <= 0 || HW <= 0 || ((uintptr_t)x & al) || ((uintptr_t)gy & al) ||
      ((uintptr_t)gx & al) || ((uintptr_t)gadd & al))
    return FMI_ERR_BAD_ARG;
  if (check_c(C)) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = rows_per_block(N, HW, 16384, 16);
  hipLaunchKernelGGL((in_bwd_apply_kernel<T>), dim3((HW + rpb - 1) / rpb, N), dim3(256), 0, st, x, gy, stats, gamma, beta, red, gx, HW, C, slope,
                     gadd, rpb);
  if (dgamma && dbeta)
    hipLaunchKernelGGL(in_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, st, red, dgamma, dbeta, N, C);
  return fmi_launch_status();
}
extern "C" int fmi_instnorm_bwd_apply_f32(const float* x, const float* gy, const float* stats, const float* gamma,
                                          const float* beta, const double* red, float* gx, float* dgamma, float* dbeta, int N,
                                          int HW, int C, float slope, void* stream) {
  return in_bwd_apply_impl<float>(x, gy, stats, gamma, beta, red, gx, dgamma, dbeta, N, HW, C, slope, nullptr, stream);
}
extern "C" int fmi_instnorm_bwd_apply_add_f32(const float* x, const float* gy, const float* stats, const float* gamma,
                                              const float* beta, const double* red, const float* gadd, float* gx, float* dgamma,
                                              float* dbeta, int N, int HW, int C, float slope, void* stream) {
  if (!gadd) return FMI_ERR_BAD_ARG;
  return in_bwd_apply_impl<float>(x, gy, stats, gamma, beta, red, gx, dgamma, dbeta, N, HW, C, slope, gadd, stream);
}
// bf16 activations (gadd may be NULL)
extern "C" int fmi_instnorm_bwd_apply_bf16(const uint16_t* x, const uint16_t* gy, const float* stats, const float* gamma, const float* beta,
                                           const double* red, const uint16_t* gadd, uint16_t* gx, float* dgamma, float* dbeta, int N, int HW,
                                           int C, float slope, void* stream) {
  return in_bwd_apply_impl<uint16_t>(x, gy, stats, gamma, beta, red, gx, dgamma, dbeta, N, HW, C, slope, gadd, stream);
}

// nn.BatchNorm2d running-statistics bookkeeping (torch/nn/modules/batchnorm.py semantics: momentum average of the batch mean and of
// the UNBIASED batch variance, num_batches_tracked += 1) from the (mean, rstd) pairs the statistics pass produced: one launch per
// BatchNorm instead of nine [C]-sized ATen kernels (104 BatchNorms per pSp step).
__global__ void __launch_bounds__(256) bn_running_update_kernel(const float* __restrict__ stats, const double* __restrict__ sums,
                                                                float* __restrict__ rmean, float* __restrict__ rvar,
                                                                int64_t* __restrict__ nbt, int C, float unbias, double count, float eps,
                                                                float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= C) return;
  const float mean = stats[2 * c], rstd = stats[2 * c + 1];
  float var;
  if (sums) {  // the fp64 (sum, sum of squares) of the statistics pass: no 1 / rstd^2 - eps cancellation when var << eps
    const double m = sums[2 * c] / count;
    double v = sums[2 * c + 1] / count - m * m;
    var = (float)((v < 0 ? 0 : v) * (double)unbias);
  } else {
    var = fmaxf(1.f / (rstd * rstd) - eps, 0.f) * unbias;
  }
  rmean[c] = __fadd_rn(__fmul_rn(rmean[c], 1.f - momentum), __fmul_rn(momentum, mean));
  rvar[c] = __fadd_rn(__fmul_rn(rvar[c], 1.f - momentum), __fmul_rn(momentum, var));
}
extern "C" int fmi_batchnorm_running_update_f32(const float* stats, const double* sums, float* running_mean, float* running_var,
                                                int64_t* num_batches_tracked, int C, int64_t count, float eps, float momentum,
                                                void* stream) {
  if (!stats || !running_mean || !running_var || C <= 0 || count <= 0) return FMI_ERR_BAD_ARG;
  const float unbias = (float)((double)count / (double)(count > 1 ? count - 1 : 1));
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, sums, running_mean, running_var,
                     num_batches_tracked, C, unbias, (double)count, eps, momentum);
  return fmi_launch_status();
}
