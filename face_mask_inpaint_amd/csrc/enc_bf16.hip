// bf16 activations of the IR-SE50 body of the pSp encoder (modules/psp/encoders/helpers.py:56-119): the element-wise kernels between
// the bf16 convolutions of conv_bf16.hip and the bf16 BatchNorm kernels of norm.hip.  NHWC bf16 tensors (uint16_t = raw bits);
// parameters (PReLU slopes), SE gates and every reduction result stay fp32.  One thread moves 8 channels (16 bytes); C % 8 == 0.
#include "common.h"

typedef uint16_t bf16_t;
__device__ __forceinline__ float e_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float e_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t e_pack(float a, float b) {  // round to nearest even
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ void e_unpack8(const uint4& v, float (&f)[8]) {
  f[0] = e_lo(v.x), f[1] = e_hi(v.x), f[2] = e_lo(v.y), f[3] = e_hi(v.y);
  f[4] = e_lo(v.z), f[5] = e_hi(v.z), f[6] = e_lo(v.w), f[7] = e_hi(v.w);
}
__device__ __forceinline__ uint4 e_pack8(const float (&f)[8]) {
  return make_uint4(e_pack(f[0], f[1]), e_pack(f[2], f[3]), e_pack(f[4], f[5]), e_pack(f[6], f[7]));
}
__device__ __forceinline__ void e_load8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
}
static bool e_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---- PReLU(C): y = x > 0 ? x : a[c] x ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prelu_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ a, uint4* __restrict__ y,
                                                         int64_t total8, int C8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    float f[8], s[8];
    e_unpack8(x[i], f);
    e_load8f(a + (i % C8) * 8, s);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : s[e] * f[e];
    y[i] = e_pack8(f);
  }
}
extern "C" int fmi_prelu_bf16(const uint16_t* x, const float* a, uint16_t* y, int64_t rows, int C, void* stream) {
  if (!x || !a || !y || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !e_al16(x) || !e_al16(y) || !e_al16(a)) return FMI_ERR_UNSUPPORTED;
  const int64_t total8 = rows * (C / 8);
  hipLaunchKernelGGL(prelu_bf16_kernel, dim3(fmi_bw_grid(total8, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, a, (uint4*)y,
                     total8, C / 8);
  return fmi_launch_status();
}
// gx = g (x > 0 ? 1 : a[c]);  ga[c] = sum_rows g x (x <= 0): every block stores its sums as one row of the partials workspace,
// a second launch adds the rows (ga WRITTEN)
__global__ void __launch_bounds__(256) prelu_bwd_bf16_kernel(const uint4* __restrict__ g, const uint4* __restrict__ x, const float* __restrict__ a,
                                                             uint4* __restrict__ gx, float* __restrict__ parts, int64_t rows, int C8,
                                                             int64_t rows_per_block) {
  __shared__ float part[256 * 8];
  const int cg = threadIdx.x % C8, rl = threadIdx.x / C8, RL = 256 / C8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float ac[8], s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    e_load8f(a + cg * 8, ac);
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      float xv[8], gv[8], o[8];
      e_unpack8(x[r * C8 + cg], xv);
      e_unpack8(g[r * C8 + cg], gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = xv[e] > 0.f ? gv[e] : ac[e] * gv[e];
        if (xv[e] <= 0.f) s[e] += gv[e] * xv[e];
      }
      gx[r * C8 + cg] = e_pack8(o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  if ((int)threadIdx.x < C8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int l = 0; l < RL; ++l) t += part[(l * C8 + threadIdx.x) * 8 + e];
      parts[((int64_t)blockIdx.x * C8 + threadIdx.x) * 8 + e] = t;
    }
  }
}
__global__ void __launch_bounds__(256) sum_rows_f32_kernel(const float* __restrict__ ws, float* __restrict__ out, int nparts, int width, float scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= width) return;
  float t = 0.f;
  for (int q = 0; q < nparts; ++q) t += ws[(int64_t)q * width + i];
  out[i] = t * scale;
}
extern "C" int fmi_prelu_bwd_bf16(const uint16_t* g, const uint16_t* x, const float* a, uint16_t* gx, float* ga, float* ws, int64_t ws_floats,
                                  int64_t rows, int C, void* stream) {
  if (!g || !x || !a || !gx || !ga || !ws || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int C8 = C / 8;
  if (C % 8 != 0 || C8 > 256 || (C8 & (C8 - 1)) != 0 || !e_al16(g) || !e_al16(x) || !e_al16(gx) || !e_al16(a) || ws_floats < C) return FMI_ERR_UNSUPPORTED;
  int64_t blocks = ceil_div64(rows, 64);
  const int64_t cap = ws_floats / C < 512 ? ws_floats / C : 512;
  if (blocks > cap) blocks = cap;
  const int64_t rpb = ceil_div64(rows, blocks);
  blocks = ceil_div64(rows, rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(prelu_bwd_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)g, (const uint4*)x, a, (uint4*)gx, ws, rows, C8, rpb);
  hipLaunchKernelGGL(sum_rows_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)ws, ga, (int)blocks, C, 1.f);
  return fmi_launch_status();
}

// ---- SE gate and residual add in one pass: y = x * s[n][c] + res ------------------------------------------------------------
__global__ void __launch_bounds__(256) scale_add_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ s, const uint4* __restrict__ res,
                                                             uint4* __restrict__ y, int64_t PC8, int C8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t n = i / PC8;
    float f[8], sc[8], r[8];
    e_unpack8(x[i], f);
    e_unpack8(res[i], r);
    e_load8f(s + (n * C8 + c8) * 8, sc);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] * sc[e] + r[e];
    y[i] = e_pack8(f);
  }
}
extern "C" int fmi_scale_channels_add_bf16(const uint16_t* x, const float* s, const uint16_t* res, uint16_t* y, int N, int64_t P, int C, void* stream) {
  if (!x || !s || !res || !y || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !e_al16(x) || !e_al16(y) || !e_al16(s) || !e_al16(res)) return FMI_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)N * P * (C / 8);
  hipLaunchKernelGGL(scale_add_bf16_kernel, dim3(fmi_bw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, s,
                     (const uint4*)res, (uint4*)y, P * (C / 8), C / 8, total);
  return fmi_launch_status();
}
// y = a + b (bf16), the residual add of a plain IR block
__global__ void __launch_bounds__(256) add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, int64_t total8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    float f[8], r[8];
    e_unpack8(a[i], f);
    e_unpack8(b[i], r);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += r[e];
    y[i] = e_pack8(f);
  }
}
extern "C" int fmi_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, int64_t n, void* stream) {
  if (!a || !b || !y || n <= 0) return FMI_ERR_BAD_ARG;
  if (n % 8 != 0 || !e_al16(a) || !e_al16(b) || !e_al16(y)) return FMI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(fmi_bw_grid(n / 8, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b, (uint4*)y, n / 8);
  return fmi_launch_status();
}

// ---- AdaptiveAvgPool2d(1) of the SE module: pooled[n][c] = mean_p x[n][p][c] (fp32 out), partial rows + one adding launch ----
__global__ void __launch_bounds__(256) gap_bf16_kernel(const uint4* __restrict__ x, float* __restrict__ parts, int64_t P, int C8, int64_t rows_per_block) {
  __shared__ float part[256 * 8];
  const int cg = threadIdx.x % C8, rl = threadIdx.x / C8, RL = 256 / C8;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL)
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      float f[8];
      e_unpack8(x[((int64_t)n * P + r) * C8 + cg], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  if ((int)threadIdx.x < C8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int l = 0; l < RL; ++l) t += part[(l * C8 + threadIdx.x) * 8 + e];
      parts[(((int64_t)n * gridDim.x + blockIdx.x) * C8 + threadIdx.x) * 8 + e] = t;
    }
  }
}
__global__ void __launch_bounds__(256) gap_finish_kernel(const float* __restrict__ parts, float* __restrict__ out, int nparts, int C, float inv) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float t = 0.f;
  for (int q = 0; q < nparts; ++q) t += parts[((int64_t)n * nparts + q) * C + c];
  out[(int64_t)n * C + c] = t * inv;
}
extern "C" int fmi_global_avgpool_bf16(const uint16_t* x, float* pooled, float* ws, int64_t ws_floats, int N, int64_t P, int C, void* stream) {
  if (!x || !pooled || !ws || N <= 0 || P <= 0 || C <= 0 || N > 65535) return FMI_ERR_BAD_ARG;
  const int C8 = C / 8;
  if (C % 8 != 0 || C8 > 256 || (C8 & (C8 - 1)) != 0 || !e_al16(x) || ws_floats < (int64_t)N * C) return FMI_ERR_UNSUPPORTED;
  int64_t blocks = ceil_div64(P, 64);
  const int64_t cap = ws_floats / ((int64_t)N * C) < 64 ? ws_floats / ((int64_t)N * C) : 64;
  if (blocks > cap) blocks = cap;
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gap_bf16_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, (const uint4*)x, ws, P, C8, rpb);
  hipLaunchKernelGGL(gap_finish_kernel, dim3((C + 255) / 256, N), dim3(256), 0, st, (const float*)ws, pooled, (int)blocks, C, 1.f / (float)P);
  return fmi_launch_status();
}
// gradient of x through BOTH consumers of the SE input: gx = g (its gradient as the scaled operand, bf16) + gpool[n][c] / P
__global__ void __launch_bounds__(256) add_bcast_bf16_kernel(const uint4* __restrict__ g, const float* __restrict__ gpool, uint4* __restrict__ gx,
                                                             int64_t PC8, int C8, int64_t total, float inv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t n = i / PC8;
    float f[8], p[8];
    e_unpack8(g[i], f);
    e_load8f(gpool + (n * C8 + c8) * 8, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += p[e] * inv;
    gx[i] = e_pack8(f);
  }
}
extern "C" int fmi_add_bcast_bf16(const uint16_t* g, const float* gpool, uint16_t* gx, int N, int64_t P, int C, void* stream) {
  if (!g || !gpool || !gx || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !e_al16(g) || !e_al16(gx) || !e_al16(gpool)) return FMI_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)N * P * (C / 8);
  hipLaunchKernelGGL(add_bcast_bf16_kernel, dim3(fmi_bw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, gpool, (uint4*)gx,
                     P * (C / 8), C / 8, total, 1.f / (float)P);
  return fmi_launch_status();
}

// fp32 twin (the fp32 IR-SE50 body of configs C3 / C5): the SE input's two gradients -- through the gated product and through the
// global average pool -- meet in one pass instead of a pool-backward pass plus an accumulation pass of the autograd engine
__global__ void __launch_bounds__(256) add_bcast_f32_kernel(const float4* __restrict__ g, const float* __restrict__ gpool, float4* __restrict__ gx,
                                                            int64_t PC4, int C4, int64_t total, float inv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t n = i / PC4;
    float4 v = g[i];
    const float4 p = *reinterpret_cast<const float4*>(gpool + (n * C4 + c4) * 4);
    v.x += p.x * inv, v.y += p.y * inv, v.z += p.z * inv, v.w += p.w * inv;
    gx[i] = v;
  }
}
extern "C" int fmi_add_bcast_f32(const float* g, const float* gpool, float* gx, int N, int64_t P, int C, void* stream) {
  if (!g || !gpool || !gx || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 != 0 || !e_al16(g) || !e_al16(gx) || !e_al16(gpool)) return FMI_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)N * P * (C / 4);
  hipLaunchKernelGGL(add_bcast_f32_kernel, dim3(fmi_bw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, gpool, (float4*)gx,
                     P * (C / 4), C / 4, total, 1.f / (float)P);
  return fmi_launch_status();
}

// ---- MaxPool2d(1, stride): y[n][oy][ox] = x[n][oy s][ox s]; backward scatters into zeros --------------------------------------
__global__ void __launch_bounds__(256) subsample_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int H, int W, int C8, int OH, int OW,
                                                             int s, int64_t total, int backward) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C8);
    int64_t r = i / C8;
    if (!backward) {  // i indexes y
      const int ox = (int)(r % OW);
      r /= OW;
      const int oy = (int)(r % OH);
      const int n = (int)(r / OH);
      y[i] = x[(((int64_t)n * H + oy * s) * W + ox * s) * C8 + c];
    } else {  // i indexes gx (= y here, of extent H x W); x is the small gradient
      const int xx = (int)(r % W);
      r /= W;
      const int yy = (int)(r % H);
      const int n = (int)(r / H);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (yy % s == 0 && xx % s == 0 && yy / s < OH && xx / s < OW) v = x[(((int64_t)n * OH + yy / s) * OW + xx / s) * C8 + c];
      y[i] = v;
    }
  }
}
extern "C" int fmi_subsample_bf16(const uint16_t* x, uint16_t* y, int N, int H, int W, int C, int stride, int backward, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !e_al16(x) || !e_al16(y)) return FMI_ERR_UNSUPPORTED;
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t total = (int64_t)N * (backward ? (int64_t)H * W : (int64_t)OH * OW) * (C / 8);
  hipLaunchKernelGGL(subsample_bf16_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, H, W, C / 8,
                     OH, OW, stride, total, backward);
  return fmi_launch_status();
}
