// bf16 activations of the StyleGAN2 decoder (configs C3 / C5): the bandwidth kernels around the bf16 convolutions of conv_bf16.hip.
// Tensors are NHWC bf16 (uint16_t = raw bits); per-sample / per-channel factors, noise maps, biases and every reduction result
// stay fp32.  One thread moves 8 channels (16 bytes) per access; C % 8 == 0 throughout.
//   scale_channels      x * s[n][c]                       ModulatedConv2d: modulation / demodulation (model.py:244-252)
//   noise_bias_act      lrelu(x + nw*noise + bias) * g    NoiseInjection + FusedLeakyReLU (model.py:282-294, op/fused_act.py:72-85)
//   upfirdn2d_nhwc      Blur after the up-convolutions    (model.py:52-68, op/upfirdn2d.py:85-147)
//   torgb               1x1 modulated conv C -> 3 without demodulation + bias + skip (model.py:349-369)
#include "common.h"
#include <cstdlib>

#ifndef FMI_HOST_EMU
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float a, float b) {  // round to nearest even
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf_lo(v.x), f[1] = bf_hi(v.x), f[2] = bf_lo(v.y), f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z), f[5] = bf_hi(v.z), f[6] = bf_lo(v.w), f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf(f[0], f[1]), pack_bf(f[2], f[3]), pack_bf(f[4], f[5]), pack_bf(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
}
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
typedef __bf16 bf16x2h __attribute__((ext_vector_type(2)));
typedef float f32x2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf_hw(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even, one instruction
  const bf16x2h r = __builtin_convertvector((f32x2h){a, b}, bf16x2h);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ uint4 pack8_hw(const float (&f)[8]) {
  return make_uint4(pack_bf_hw(f[0], f[1]), pack_bf_hw(f[2], f[3]), pack_bf_hw(f[4], f[5]), pack_bf_hw(f[6], f[7]));
}

// ---------------------------------------------------------------------------------------------------------------------------
// y[n][p][c] = x[n][p][c] * s[n][c]
__global__ void __launch_bounds__(256) scale_channels_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ s,
                                                                  uint4* __restrict__ y, int64_t PC8, int C8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t n = i / PC8;
    float f[8], sc[8];
    unpack8(x[i], f);
    load8f(s + (n * C8 + c8) * 8, sc);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= sc[e];
    y[i] = pack8(f);
  }
}
extern "C" int fmi_scale_channels_bf16(const uint16_t* x, const float* s, uint16_t* y, int N, int64_t P, int C, void* stream) {
  if (!x || !s || !y || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !al16(x) || !al16(y) || !al16(s)) return FMI_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)N * P * (C / 8);
  hipLaunchKernelGGL(scale_channels_bf16_kernel, dim3(fmi_bw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, s, (uint4*)y, P * (C / 8), C / 8, total);
  return fmi_launch_status();
}

// Shared shape of the per-channel reductions below: 256 threads = (256 / C8) rows x C8 channel chunks per pass; the threads that
// own one chunk are combined through LDS.  The block's sums are STORED as row `dst` of a partials workspace: thousands of fp32 atomics on a few hundred
// addresses serialise in L2 (measured: 0.2 ms per launch for 2048 blocks x 128 channels, 4x the streaming time of the pass itself);
// sum_parts_kernel adds the rows afterwards -- deterministic, and nothing has to be zeroed.
__device__ __forceinline__ void chunk_reduce_store(float (&acc)[8], float* __restrict__ dst /*[C]*/, int C8, float (*part)[8]) {
  const int RL = 256 / C8, cg = threadIdx.x % C8;
#pragma unroll
  for (int e = 0; e < 8; ++e) part[threadIdx.x][e] = acc[e];
  __syncthreads();
  if ((int)threadIdx.x < C8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = part[threadIdx.x][e];
    for (int l = 1; l < RL; ++l)
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += part[l * C8 + cg][e];
    *reinterpret_cast<float4*>(dst + 8 * cg) = make_float4(t[0], t[1], t[2], t[3]);
    *reinterpret_cast<float4*>(dst + 8 * cg + 4) = make_float4(t[4], t[5], t[6], t[7]);
  }
  __syncthreads();
}
// out[g][i] = sum_{p < nparts} ws[(g * nparts + p) * width + i]
// 64 columns x 16 row lanes per workgroup, eight independent loads in flight per thread, the lanes' sums added in a fixed order (one
// thread per column walking up to 512 rows with four accumulators was a latency chain: 18 us per launch, 43 launches per C5 step)
__global__ void __launch_bounds__(1024) sum_parts_kernel(const float* __restrict__ ws, float* __restrict__ out, int nparts, int width) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const float* p = ws + (int64_t)blockIdx.y * nparts * width + i;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += part[l][tx];
    out[(int64_t)blockIdx.y * width + i] = t;
  }
}
static void launch_sum_parts(const float* ws, float* out, int nparts, int width, int64_t groups, hipStream_t st) {
  hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((width + 63) / 64), (unsigned)groups), dim3(1024), 0, st, ws, out, nparts, width);
}
// how many row blocks a reduction pass uses: enough to fill the chip, at most what the workspace holds (width floats per block)
static int64_t parts_for(int64_t rows, int64_t groups, int64_t width, int64_t ws_floats) {
  int64_t b = ceil_div64(rows, 64);
  const int64_t want = ceil_div64(2048, groups);
  if (b > want) b = want;
  const int64_t fit = ws_floats / (groups * width);
  if (b > fit) b = fit;
  return b;
}

// gs[n][c] = sum_p g[n][p][c] * x[n][p][c]   (written, not accumulated; ws: partials workspace of >= N*C floats)
__global__ void __launch_bounds__(256) scale_channels_gs_bf16_kernel(const uint4* __restrict__ g, const uint4* __restrict__ x,
                                                                     float* __restrict__ ws, int64_t P, int C8, int64_t rows_per_block) {
  __shared__ float part[256][8];
  const int RL = 256 / C8, cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const uint4* gb = g + (int64_t)n * P * C8;
  const uint4* xb = x + (int64_t)n * P * C8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {  // two independent pairs of 16-byte loads in flight
    float a[8], b[8], a2[8], b2[8];
    const uint4 ga = gb[r * C8 + cg], xa = xb[r * C8 + cg], gc = gb[(r + RL) * C8 + cg], xc = xb[(r + RL) * C8 + cg];
    unpack8(ga, a), unpack8(xa, b), unpack8(gc, a2), unpack8(xc, b2);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(a2[e], b2[e], fmaf(a[e], b[e], acc[e]));
  }
  for (; r < r1; r += RL) {
    float a[8], b[8];
    unpack8(gb[r * C8 + cg], a);
    unpack8(xb[r * C8 + cg], b);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(a[e], b[e], acc[e]);
  }
  chunk_reduce_store(acc, ws + ((int64_t)n * gridDim.x + blockIdx.x) * C8 * 8, C8, part);
}
static bool c8_ok(int C) {
  const int c8 = C / 8;
  return C % 8 == 0 && c8 >= 1 && c8 <= 256 && (c8 & (c8 - 1)) == 0;
}
extern "C" int fmi_scale_channels_gs_bf16(const uint16_t* g, const uint16_t* x, float* gs, float* ws, int64_t ws_floats, int N, int64_t P,
                                          int C, void* stream) {
  if (!g || !x || !gs || !ws || N <= 0 || P <= 0 || C <= 0 || N > 65535 || ws_floats < (int64_t)N * C) return FMI_ERR_BAD_ARG;
  if (!c8_ok(C) || !al16(g) || !al16(x) || !al16(ws)) return FMI_ERR_UNSUPPORTED;
  int64_t blocks = parts_for(P, N, C, ws_floats);
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipLaunchKernelGGL(scale_channels_gs_bf16_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, (const uint4*)g,
                     (const uint4*)x, ws, P, C / 8, rpb);
  launch_sum_parts(ws, gs, (int)blocks, C, N, (hipStream_t)stream);
  return fmi_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// y = lrelu(x + bias[c] + nw * noise[p], alpha) * scale
__global__ void __launch_bounds__(256) noise_bias_act_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ bias,
                                                                  const float* __restrict__ noise, const float* __restrict__ nw,
                                                                  uint4* __restrict__ y, int64_t total, int C8, float alpha, float scale) {
  const float w = (noise && nw) ? nw[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C8;
    const int c8 = (int)(i - p * C8);
    float f[8], b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unpack8(x[i], f);
    if (bias) load8f(bias + 8 * c8, b);
    const float nz = (noise && nw) ? w * noise[p] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = f[e] + b[e] + nz;
      f[e] = (v > 0.f ? v : v * alpha) * scale;
    }
    y[i] = pack8(f);
  }
}
extern "C" int fmi_noise_bias_act_bf16(const uint16_t* x, const float* bias, const float* noise, const float* nw, uint16_t* y,
                                       int64_t pixels, int C, float alpha, float scale, void* stream) {
  if (!x || !y || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 8 != 0 || !al16(x) || !al16(y) || !al16(bias)) return FMI_ERR_UNSUPPORTED;
  const int64_t total = pixels * (C / 8);
  hipLaunchKernelGGL(noise_bias_act_bf16_kernel, dim3(fmi_bw_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     bias, noise, nw, (uint4*)y, total, C / 8, alpha, scale);
  return fmi_launch_status();
}

// backward: gx = g * scale * (y > 0 ? 1 : alpha);  gbias[c] = sum_p gx;  gnw = sum gx * noise[p]   (one pass; both written).
// ws: partials workspace, rows of C + 8 floats (bias sums, then the noise-weight sum), >= C + 8 floats
__global__ void __launch_bounds__(256) noise_bias_act_bwd_bf16_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y,
                                                                      const float* __restrict__ noise, uint4* __restrict__ gx,
                                                                      float* __restrict__ ws, int want_nw, int64_t P,
                                                                      int C8, float alpha, float scale, int64_t rows_per_block) {
  __shared__ float part[256][8];
  __shared__ float red[4];
  const int RL = 256 / C8, cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float an = 0.f;
  auto one = [&](int64_t r, const uint4& gv, const uint4& yv) {
    float a[8], b[8];
    unpack8(gv, a);
    unpack8(yv, b);
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = a[e] * scale * (b[e] > 0.f ? 1.f : alpha);
      acc[e] += a[e];
      rs += a[e];
    }
    if (noise) an = fmaf(rs, noise[r], an);
    gx[r * C8 + cg] = pack8(a);
  };
  int64_t r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {  // two independent pairs of 16-byte loads in flight
    const uint4 g0 = g[r * C8 + cg], y0 = y[r * C8 + cg], g1 = g[(r + RL) * C8 + cg], y1 = y[(r + RL) * C8 + cg];
    one(r, g0, y0);
    one(r + RL, g1, y1);
  }
  for (; r < r1; r += RL) one(r, g[r * C8 + cg], y[r * C8 + cg]);
  const int W = C8 * 8 + 8;
  chunk_reduce_store(acc, ws + (int64_t)blockIdx.x * W, C8, part);
  if (want_nw) {
    an = block_sum_256(an, red);
    if (threadIdx.x == 0) ws[(int64_t)blockIdx.x * W + C8 * 8] = an;
  }
}
// gbias[c] = sum of column c, gnw = sum of column C of the partial rows.  64 columns x 16 row lanes per block, 8 independent loads in
// flight per thread (one thread per column walking 2048 rows took 0.2 ms: pure load latency)
__global__ void __launch_bounds__(1024) nba_finish_kernel(const float* __restrict__ ws, float* __restrict__ gbias, float* __restrict__ gnw,
                                                          int nparts, int C) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx, W = C + 8;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i <= C) {
    int q = ty;
    for (; q + 7 * 16 < nparts; q += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(int64_t)(q + 16 * u) * W + i];
    }
    for (; q < nparts; q += 16) a[0] += ws[(int64_t)q * W + i];
  }
  part[ty][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (ty == 0 && i <= C) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += part[l][tx];
    if (i < C) {
      if (gbias) gbias[i] = t;
    } else if (gnw) {
      gnw[0] = t;
    }
  }
}
extern "C" int fmi_noise_bias_act_bwd_bf16(const uint16_t* g, const uint16_t* y, const float* noise, uint16_t* gx, float* gnw,
                                           float* gbias, float* ws, int64_t ws_floats, int64_t pixels, int C, float alpha, float scale,
                                           void* stream) {
  if (!g || !y || !gx || !ws || pixels <= 0 || C <= 0 || ws_floats < C + 8) return FMI_ERR_BAD_ARG;
  if (!c8_ok(C) || !al16(g) || !al16(y) || !al16(gx) || !al16(ws)) return FMI_ERR_UNSUPPORTED;
  int64_t blocks = parts_for(pixels, 1, C + 8, ws_floats);
  const int64_t rpb = ceil_div64(pixels, blocks);
  blocks = ceil_div64(pixels, rpb);
  const int want_nw = (noise && gnw) ? 1 : 0;
  hipLaunchKernelGGL(noise_bias_act_bwd_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)g,
                     (const uint4*)y, want_nw ? noise : nullptr, (uint4*)gx, ws, want_nw, pixels, C / 8, alpha, scale, rpb);
  if (gbias || want_nw)
    hipLaunchKernelGGL(nba_finish_kernel, dim3((C + 1 + 63) / 64), dim3(1024), 0, (hipStream_t)stream, ws, gbias, want_nw ? gnw : nullptr,
                       (int)blocks, C);
  return fmi_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Adjoint of the fused StyledConv output stage  y = lrelu(z * d[n][c] + nw * noise[n][p] + bias[c], slope) * gain  (z: the Blur's /
// the convolution's fp32 result; fmi_blur_act_bf16, fmi_conv2d_fwd_act_bf16) in ONE pass over g and y:
//   gpre = g * gain * (y > 0 ? 1 : slope)                       t[n][p][c] = gpre * d[n][c]   (gradient wrt z, bf16)
//   gbias[c] = sum_{n,p} gpre        gnw = sum gpre * noise     gd[n][c] = sum_p gpre * z
// z itself was never stored: gpre * (z d + nw noise + bias) = g * y exactly (both branches of the leaky ReLU), so
//   gd[n][c] = (sum_p g y - nw sum_p gpre noise - bias[c] sum_p gpre) / d[n][c].
// The unfused chain was three passes (noise_bias_act_bwd, scale_channels, scale_channels_gs: 4 reads and 2 writes of the map).
// ws: per (sample, row block) three rows of C floats: sum g y, sum gpre noise, sum gpre.
__global__ void __launch_bounds__(256) styled_out_bwd_bf16_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y,
                                                                  const float* __restrict__ noise, const float* __restrict__ colscale,
                                                                  uint4* __restrict__ t, float* __restrict__ ws, int64_t P, int C8,
                                                                  float slope, float gain, int64_t rows_per_block) {
  __shared__ float part[256][8];
  const int RL = 256 / C8, cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const uint4* gb = g + (int64_t)n * P * C8;
  const uint4* yb = y + (int64_t)n * P * C8;
  uint4* tb = t + (int64_t)n * P * C8;
  const float* nb = noise ? noise + (int64_t)n * P : nullptr;
  float d[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] = 1.f;
  if (colscale) load8f(colscale + ((int64_t)n * C8 + cg) * 8, d);
  float sa[8], sb[8], sc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sa[e] = 0.f, sb[e] = 0.f, sc[e] = 0.f;
  auto one = [&](int64_t r, const uint4& gv, const uint4& yv) {
    float a[8], b[8];
    unpack8(gv, a);
    unpack8(yv, b);
    const float nz = nb ? nb[r] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sa[e] = fmaf(a[e], b[e], sa[e]);
      const float gp = a[e] * gain * (b[e] > 0.f ? 1.f : slope);
      sb[e] = fmaf(gp, nz, sb[e]);
      sc[e] += gp;
      a[e] = gp * d[e];
    }
    tb[r * C8 + cg] = pack8_hw(a);
  };
  int64_t r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {  // two independent pairs of 16-byte loads in flight
    const uint4 g0 = gb[r * C8 + cg], y0 = yb[r * C8 + cg], g1 = gb[(r + RL) * C8 + cg], y1 = yb[(r + RL) * C8 + cg];
    one(r, g0, y0);
    one(r + RL, g1, y1);
  }
  for (; r < r1; r += RL) one(r, gb[r * C8 + cg], yb[r * C8 + cg]);
  float* row = ws + ((int64_t)n * gridDim.x + blockIdx.x) * 3 * C8 * 8;
  chunk_reduce_store(sa, row, C8, part);
  chunk_reduce_store(sb, row + C8 * 8, C8, part);
  chunk_reduce_store(sc, row + 2 * C8 * 8, C8, part);
}
// sums[n][3][C] -> gd[n][c], gbias[c], gnw (one block; N * C is a few thousand)
__global__ void __launch_bounds__(256) styled_out_finish_kernel(const float* __restrict__ sums, const float* __restrict__ colscale,
                                                                const float* __restrict__ nw, const float* __restrict__ bias,
                                                                float* __restrict__ gd, float* __restrict__ gbias, float* __restrict__ gnw,
                                                                int N, int C) {
  __shared__ float red[4];
  const float nwv = nw ? nw[0] : 0.f;
  float tn = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float bv = bias ? bias[c] : 0.f;
    float tb = 0.f;
    for (int n = 0; n < N; ++n) {
      const float* s = sums + (int64_t)n * 3 * C;
      const float A = s[c], B = s[C + c], Cs = s[2 * C + c];
      if (gd) gd[(int64_t)n * C + c] = (A - nwv * B - bv * Cs) / (colscale ? colscale[(int64_t)n * C + c] : 1.f);
      tb += Cs;
      tn += B;
    }
    if (gbias) gbias[c] = tb;
  }
  tn = block_sum_256(tn, red);
  if (gnw && threadIdx.x == 0) gnw[0] = tn;
}
/* t = gradient wrt the pre-demodulation map (bf16), gd [N][C], gbias [C], gnw [1] (each of the three may be NULL); noise, colscale, nw,
 * bias as the forward was given them (NULL where it had none).  ws: >= N * 3 * C floats (more = more row blocks); sums: N * 3 * C floats. */
extern "C" int fmi_styled_out_bwd_bf16(const uint16_t* g, const uint16_t* y, const float* noise, const float* colscale, const float* nw,
                                       const float* bias, uint16_t* t, float* gd, float* gbias, float* gnw, float* ws, int64_t ws_floats,
                                       float* sums, int N, int64_t P, int C, float slope, float gain, void* stream) {
  if (!g || !y || !t || !ws || !sums || N <= 0 || P <= 0 || C <= 0 || (noise && !nw)) return FMI_ERR_BAD_ARG;
  if (!c8_ok(C) || !al16(g) || !al16(y) || !al16(t) || !al16(ws) || (colscale && !al16(colscale))) return FMI_ERR_UNSUPPORTED;
  if (ws_floats < (int64_t)N * 3 * C) return FMI_ERR_BAD_ARG;
  int64_t blocks = parts_for(P, N, 3 * (int64_t)C, ws_floats);
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(styled_out_bwd_bf16_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, st, (const uint4*)g, (const uint4*)y, noise,
                     colscale, (uint4*)t, ws, P, C / 8, slope, gain, rpb);
  launch_sum_parts(ws, sums, (int)blocks, 3 * C, N, st);
  if (gd || gbias || gnw)
    hipLaunchKernelGGL(styled_out_finish_kernel, dim3(1), dim3(256), 0, st, sums, colscale, noise ? nw : nullptr, bias, gd, gbias,
                       noise ? gnw : nullptr, N, C);
  return fmi_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// upfirdn2d on NHWC bf16, FIR form (up = down = 1, square kernel of 2..4 taps: the decoder's Blur and its gradient).  One thread =
// (2 adjacent output pixels, 8 channels); the KH x (KW+1) window is read once with 16-byte loads; fp32 taps and accumulation.
template <int KH, int KW>
__global__ void __launch_bounds__(256) upfirdn2d_nhwc_fir_bf16_kernel(const uint4* __restrict__ in, const float* __restrict__ kernel,
                                                                      uint4* __restrict__ out, int in_h, int in_w, int C8, int out_h,
                                                                      int out_w, int pad_x0, int pad_y0, int total) {
  float kf[KH][KW];
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int b = 0; b < KW; ++b) kf[a][b] = kernel[(KH - 1 - a) * KW + (KW - 1 - b)];
  const int pw = (out_w + 1) >> 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c8 = i % C8;
    int r = i / C8;
    const int px = r % pw;
    r /= pw;
    const int oy = r % out_h, n = r / out_h;
    const int ox = 2 * px;
    const int iy0 = oy - pad_y0, ix0 = ox - pad_x0;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = 0.f, a1[e] = 0.f;
    const uint4* base = in + (int64_t)n * in_h * in_w * C8 + c8;
#pragma unroll
    for (int a = 0; a < KH; ++a) {
      const int iy = iy0 + a;
      if ((unsigned)iy >= (unsigned)in_h) continue;
      const uint4* row = base + (int64_t)iy * in_w * C8;
#pragma unroll
      for (int b = 0; b <= KW; ++b) {
        const int ix = ix0 + b;
        if ((unsigned)ix >= (unsigned)in_w) continue;
        float v[8];
        unpack8(row[(int64_t)ix * C8], v);
        if (b < KW) {
          const float k0 = kf[a][b];
#pragma unroll
          for (int e = 0; e < 8; ++e) a0[e] = fmaf(v[e], k0, a0[e]);
        }
        if (b > 0) {
          const float k1 = kf[a][b - 1];
#pragma unroll
          for (int e = 0; e < 8; ++e) a1[e] = fmaf(v[e], k1, a1[e]);
        }
      }
    }
    uint4* o = out + ((int64_t)(n * out_h + oy) * out_w + ox) * C8 + c8;
    o[0] = pack8(a0);
    if (ox + 1 < out_w) o[C8] = pack8(a1);
  }
}
// Column-run form of the same FIR, bf16 (E = 8 channels per 16 bytes) and fp32 (E = 4): one thread = (2 adjacent output columns,
// R consecutive output rows, E channels).  Every input row of the run is read ONCE as KW+1 16-byte loads and feeds up to KH output
// rows: (R + KH - 1)(KW + 1) loads for 2R outputs -- 4.4 per output at R = 4 against 10 for the per-pixel-pair form, which was bound
// by load issue, not by HBM (bf16 moved half the bytes in the same time as fp32).
template <int KH, int KW, int R, bool BF>
__global__ void __launch_bounds__(256) upfirdn2d_nhwc_fir_run_kernel(const uint4* __restrict__ in, const float* __restrict__ kernel,
                                                                     uint4* __restrict__ out, int in_h, int in_w, int CV, int out_h,
                                                                     int out_w, int pad_x0, int pad_y0, int total) {
  constexpr int E = BF ? 8 : 4;
  float kf[KH][KW];
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int b = 0; b < KW; ++b) kf[a][b] = kernel[(KH - 1 - a) * KW + (KW - 1 - b)];
  const int pw = (out_w + 1) >> 1, ph = (out_h + R - 1) / R;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int cv = i % CV;
    int r = i / CV;
    const int px = r % pw;
    r /= pw;
    const int ry = r % ph, n = r / ph;
    const int ox = 2 * px, oy0 = ry * R;
    const int iy0 = oy0 - pad_y0, ix0 = ox - pad_x0;
    float acc[R][2][E];
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[q][0][e] = 0.f, acc[q][1][e] = 0.f;
    const uint4* base = in + (int64_t)n * in_h * in_w * CV + cv;
#pragma unroll
    for (int t = 0; t < R + KH - 1; ++t) {
      const int iy = iy0 + t;
      if ((unsigned)iy >= (unsigned)in_h) continue;
      const uint4* row = base + (int64_t)iy * in_w * CV;
      float v[KW + 1][E];
#pragma unroll
      for (int b = 0; b <= KW; ++b) {
        const int ix = ix0 + b;
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)ix < (unsigned)in_w) raw = row[(int64_t)ix * CV];
        if constexpr (BF) {
          unpack8(raw, v[b]);
        } else {
          v[b][0] = __uint_as_float(raw.x), v[b][1] = __uint_as_float(raw.y), v[b][2] = __uint_as_float(raw.z), v[b][3] = __uint_as_float(raw.w);
        }
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int a = t - q;  // tap row of this input row for output row q (compile-time after unrolling)
        if (a < 0 || a >= KH) continue;
#pragma unroll
        for (int b = 0; b < KW; ++b) {
          const float k0 = kf[a][b];
#pragma unroll
          for (int e = 0; e < E; ++e) {
            acc[q][0][e] = fmaf(v[b][e], k0, acc[q][0][e]);
            acc[q][1][e] = fmaf(v[b + 1][e], k0, acc[q][1][e]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int oy = oy0 + q;
      if (oy >= out_h) break;
      uint4* o = out + ((int64_t)(n * out_h + oy) * out_w + ox) * CV + cv;
      if constexpr (BF) {
        o[0] = pack8(acc[q][0]);
        if (ox + 1 < out_w) o[CV] = pack8(acc[q][1]);
      } else {
        o[0] = make_uint4(__float_as_uint(acc[q][0][0]), __float_as_uint(acc[q][0][1]), __float_as_uint(acc[q][0][2]), __float_as_uint(acc[q][0][3]));
        if (ox + 1 < out_w)
          o[CV] = make_uint4(__float_as_uint(acc[q][1][0]), __float_as_uint(acc[q][1][1]), __float_as_uint(acc[q][1][2]), __float_as_uint(acc[q][1][3]));
      }
    }
  }
}
// shared launcher (also called by the fp32 entry in sg2.hip): CV = channels / E vectors per pixel; returns false if it does not apply
extern "C" int fmi_internal_fir_run_launch(const void* in, const float* kernel, void* out, int N, int in_h, int in_w, int CV, int out_h,
                                           int out_w, int k, int pad_x0, int pad_y0, int bf16, void* stream) {
  // rows per thread, measured (GB/s of in + out at the decoder's Blur shapes): fp32 R = 8: 4.0-4.9 TB/s (R = 4: 3.5-4.3); bf16 R = 4: 3.7-4.2
  // (R = 8: 3.4-3.9, 128 accumulators cost occupancy); the per-pixel-pair form: 2.4 / 1.9-2.4
  const int R = bf16 ? 4 : 8;
  const int64_t tv = (int64_t)N * ((out_h + R - 1) / R) * ((out_w + 1) / 2) * CV;
  if (tv >= (1ll << 31) || out_h < 2 * R) return 0;
  static const bool off = getenv("FMI_FIR_RUN_OFF") != nullptr;
  if (off) return 0;
  const int grid = fmi_bw_grid(tv, 256);
  hipStream_t st = (hipStream_t)stream;
#define RUN_LAUNCH(K_, BF_)                                                                                                          \
  hipLaunchKernelGGL((upfirdn2d_nhwc_fir_run_kernel<K_, K_, (BF_ ? 4 : 8), BF_>), dim3(grid), dim3(256), 0, st, (const uint4*)in, kernel, \
                     (uint4*)out, in_h, in_w, CV, out_h, out_w, pad_x0, pad_y0, (int)tv)
  if (bf16) {
    if (k == 4) RUN_LAUNCH(4, true);
    else if (k == 3) RUN_LAUNCH(3, true);
    else RUN_LAUNCH(2, true);
  } else {
    if (k == 4) RUN_LAUNCH(4, false);
    else if (k == 3) RUN_LAUNCH(3, false);
    else RUN_LAUNCH(2, false);
  }
#undef RUN_LAUNCH
  return 1;
}
// ---------------------------------------------------------------------------------------------------------------------------
// SEPARABLE 4 x 4 FIR (the decoder's Blur and its gradient: the taps are an outer product by construction, model.py:36-45) through
// LDS, with the rest of a StyledConv's output folded in when asked:
//   out = lrelu(FIR(in) * colscale[n][c] + nw * noise[n][oy][ox] + bias[c], slope) * gain
// (stylegan2/model.py:88-91 Blur, :250-252 demodulation, :282-294 NoiseInjection, op/fused_act.py:30-37 FusedLeakyReLU) -- the
// blurred map, the demodulated map and the noise + bias + activation pass each were a read and a write of the whole activation.
// What bounds a FIR on 8-channel bf16 vectors is instruction issue, not HBM: at 5 TB/s of in + out a CU has ~250 vector instructions
// per 16 output bytes, and the column-run kernel above spends ~165 (16 taps x 8 channels, 4.4 global loads with bounds checks and 4.4
// unpacks per output, a five-instruction software rounding per value).  Here:
//  * a workgroup owns TH x 29 output pixels x CVT channel vectors; its (TH + 3) x 32 input pixels arrive ONCE by LDS-DMA, one
//    instruction per image row (32 pixels x 128 bytes = the workgroup's 256 lanes x 16 bytes): the per-lane part of the address is a
//    constant, a row costs an add and a select;
//  * a thread keeps 4 rows x 2 columns of outputs and walks the seven input rows: five ds_read_b128 and five unpacks per row feed
//    two row sums (4 taps each), which feed up to four output rows (1 tap each): 8 + 4 multiplies per output instead of 16;
//  * pixels p and p ^ 1 trade places in LDS where bit S of p is set (source-side, the DMA writes lane-linear): the even pixels a
//    16-lane read group touches then alternate between the two halves of a 256-byte bank row;
//  * v_cvt_pk_bf16_f32 rounds the outputs.
static __device__ __attribute__((aligned(16))) float blur_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};
struct BlurActArgs {
  const uint4* in;
  uint4* out;
  const float* kernel;    // [4][4], rank one
  const float* colscale;  // [N][C] or null
  const float* noise;     // [N][OH][OW] or null
  const float* nw;        // one float, used with noise
  const float* bias;      // [C] or null
  float slope, gain;
  int act;                // 0: plain FIR
  int N, in_h, in_w, CV, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, cgroups;
};
template <int CVT, bool ACT>
__global__ void __launch_bounds__(256) blur4_lds_bf16_kernel(BlurActArgs a) {
  constexpr int TWO = 29, R = 4, NRG = 256 / (CVT * 16), TH = R * NRG, IHT = TH + 3;
  constexpr int RPI = 8 / CVT;                 // image rows per DMA instruction (32 pixels x CVT vectors each)
  constexpr int NLD = (IHT + RPI - 1) / RPI;   // DMA instructions per thread
  constexpr int S = CVT == 8 ? 1 : 2;          // swizzle bit
  __shared__ uint4 tile[NLD * 256];
  const int tid = threadIdx.x;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = b % a.cgroups;
  b /= a.cgroups;
  const int tx = b % a.tiles_x;
  b /= a.tiles_x;
  const int ty = b % a.tiles_y, n = b / a.tiles_y;
  const int ox0 = tx * TWO, oy0 = ty * TH, cv0 = cg * CVT;
  // ---- input tile: LDS position (row r, pixel slot ps, vector cv) holds image pixel ox0 - pad_x0 + (ps ^ ((ps >> S) & 1))
  {
    const int cvl = tid % CVT, ps = (tid / CVT) & 31, rsub = tid / (CVT * 32);
    const int gx = ox0 - a.pad_x0 + (ps ^ ((ps >> S) & 1));
    const bool xok = (unsigned)gx < (unsigned)a.in_w;
    const uint4* col = a.in + (int64_t)n * a.in_h * a.in_w * a.CV + (int64_t)gx * a.CV + cv0 + cvl;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)tile;
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(tid >> 6) * 1024u);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int gy = oy0 - a.pad_y0 + j * RPI + rsub;
      const bool ok = xok && (unsigned)gy < (unsigned)a.in_h && j * RPI + rsub < IHT;
      const void* src = ok ? (const void*)(col + (int64_t)gy * a.in_w * a.CV) : (const void*)blur_zero_chunk;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(wbase + (uint32_t)j * 4096u)
                   : "memory");
    }
  }
  float kx[4], ky[4];  // k[i][j] (flipped, as upfirdn2d applies it) = ky[i] * kx[j]
  {
    const float k00 = a.kernel[15], inv = 1.f / k00;
#pragma unroll
    for (int j = 0; j < 4; ++j) kx[j] = a.kernel[15 - j] * inv, ky[j] = a.kernel[(3 - j) * 4 + 3];
  }
  const int cv = tid % CVT, xp = (tid / CVT) & 15, rg = tid / (CVT * 16);
  // LDS vector index of pixel slot for image-order pixel p of the tile row: (p ^ ((p >> S) & 1)) * CVT + cv
  int slot[5];
#pragma unroll
  for (int bb = 0; bb < 5; ++bb) {
    const int p = 2 * xp + bb;  // <= 34: the last pair's columns lie outside the 29 outputs; clamp keeps the read inside the row
    const int pc = p < 32 ? p : 31;
    slot[bb] = (pc ^ ((pc >> S) & 1)) * CVT + cv;
  }
  float acc[R][2][8];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][0][e] = 0.f, acc[q][1][e] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < R + 3; ++t) {
    const uint4* row = tile + (rg * R + t) * 32 * CVT;
    float h0[8], h1[8];
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) {
      float v[8];
      unpack8(row[slot[bb]], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (bb < 4) h0[e] = bb == 0 ? v[e] * kx[0] : fmaf(v[e], kx[bb], h0[e]);
        if (bb > 0) h1[e] = bb == 1 ? v[e] * kx[0] : fmaf(v[e], kx[bb - 1], h1[e]);
      }
    }
    // the next row's LDS reads stay behind this row's sums (a bare memory clobber does not do it: nothing ties it to the
    // arithmetic, and with all 35 vectors read up front the kernel needed 250 registers)
    asm volatile(""
                 : "+v"(h0[0]), "+v"(h0[1]), "+v"(h0[2]), "+v"(h0[3]), "+v"(h0[4]), "+v"(h0[5]), "+v"(h0[6]), "+v"(h0[7]), "+v"(h1[0]), "+v"(h1[1]),
                   "+v"(h1[2]), "+v"(h1[3]), "+v"(h1[4]), "+v"(h1[5]), "+v"(h1[6]), "+v"(h1[7])::"memory");
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int ta = t - q;  // tap row (compile-time after unrolling)
      if (ta < 0 || ta >= 4) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[q][0][e] = fmaf(h0[e], ky[ta], acc[q][0][e]), acc[q][1][e] = fmaf(h1[e], ky[ta], acc[q][1][e]);
    }
  }
  const int oxa = ox0 + 2 * xp;
  const int xlim = ox0 + TWO < a.out_w ? ox0 + TWO : a.out_w;
  if (oxa >= xlim) return;
  const bool two = oxa + 1 < xlim;
  const int c = (cv0 + cv) * 8;
  float d[8], bs[8];
  if (ACT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = 1.f, bs[e] = 0.f;
    if (a.colscale) load8f(a.colscale + (int64_t)n * a.CV * 8 + c, d);
    if (a.bias) load8f(a.bias + c, bs);
  }
  const float nwv = (ACT && a.noise) ? a.nw[0] : 0.f;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int oy = oy0 + rg * R + q;
    if (oy >= a.out_h) break;
    const int64_t pix = ((int64_t)n * a.out_h + oy) * a.out_w + oxa;
    if (ACT) {
      float nz0 = 0.f, nz1 = 0.f;
      if (a.noise) nz0 = nwv * a.noise[pix], nz1 = two ? nwv * a.noise[pix + 1] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float u0 = acc[q][0][e] * d[e] + nz0 + bs[e], u1 = acc[q][1][e] * d[e] + nz1 + bs[e];
        acc[q][0][e] = (u0 < 0.f ? u0 * a.slope : u0) * a.gain;
        acc[q][1][e] = (u1 < 0.f ? u1 * a.slope : u1) * a.gain;
      }
    }
    uint4* o = a.out + pix * a.CV + cv0 + cv;
    o[0] = pack8_hw(acc[q][0]);
    if (two) o[a.CV] = pack8_hw(acc[q][1]);
  }
}
extern "C" int fmi_upfirdn2d_nhwc_bf16(const uint16_t* in, const float* kernel, uint16_t* out, int N, int in_h, int in_w, int C, int kh,
                                       int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                       void* stream);
static int blur4_lds_launch(BlurActArgs& a, int C, hipStream_t st) {
  const bool wide = C % 64 == 0;
  const int cvt = wide ? 8 : 4, th = wide ? 8 : 16;
  a.CV = C / 8;
  a.cgroups = a.CV / cvt;
  a.tiles_x = (a.out_w + 28) / 29;
  a.tiles_y = (a.out_h + th - 1) / th;
  const int64_t nwg = (int64_t)a.N * a.tiles_y * a.tiles_x * a.cgroups;
  if (nwg > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
#define BLUR_GO(CVT_, ACT_) hipLaunchKernelGGL((blur4_lds_bf16_kernel<CVT_, ACT_>), dim3((unsigned)nwg), dim3(256), 0, st, a)
  if (wide) {
    if (a.act) BLUR_GO(8, true);
    else BLUR_GO(8, false);
  } else {
    if (a.act) BLUR_GO(4, true);
    else BLUR_GO(4, false);
  }
#undef BLUR_GO
  return fmi_launch_status();
}
/* out = lrelu(FIR4x4(in) * colscale[n][c] + nw * noise[n][oy][ox] + bias[c], slope) * gain on bf16 NHWC maps: the Blur of an
 * upsampling StyledConv with the demodulation, NoiseInjection and FusedLeakyReLU that follow it (any of colscale / noise / bias may be
 * null; nw is required with noise).  kernel: 16 fp32 taps (upfirdn2d's orientation). */
extern "C" int fmi_blur_act_bf16(const uint16_t* in, const float* kernel, uint16_t* out, int N, int in_h, int in_w, int C, int pad_x0,
                                 int pad_x1, int pad_y0, int pad_y1, const float* colscale, const float* noise, const float* nw,
                                 const float* bias, float slope, float gain, int separable, void* stream) {
  if (!in || !kernel || !out || N <= 0 || in_h <= 0 || in_w <= 0 || C <= 0 || (noise && !nw)) return FMI_ERR_BAD_ARG;
  const int out_h = in_h + pad_y0 + pad_y1 - 3, out_w = in_w + pad_x0 + pad_x1 - 3;
  if (out_h <= 0 || out_w <= 0) return FMI_ERR_BAD_ARG;
  if (C % 32 != 0 || !al16(in) || !al16(out) || (colscale && !al16(colscale)) || (bias && !al16(bias))) return FMI_ERR_UNSUPPORTED;
  BlurActArgs a{};
  a.in = (const uint4*)in; a.out = (uint4*)out; a.kernel = kernel;
  a.colscale = colscale; a.noise = noise; a.nw = nw; a.bias = bias; a.slope = slope; a.gain = gain; a.act = (colscale || noise || bias || slope != 1.f || gain != 1.f) ? 1 : 0;
  if (!separable) {  // general taps: the plain FIR through fmi_upfirdn2d_nhwc_bf16's kernels, no fused form
    if (a.act) return FMI_ERR_UNSUPPORTED;
    return fmi_upfirdn2d_nhwc_bf16(in, kernel, out, N, in_h, in_w, C, 4, 4, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, stream);
  }
  a.N = N; a.in_h = in_h; a.in_w = in_w; a.out_h = out_h; a.out_w = out_w; a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
  return blur4_lds_launch(a, C, (hipStream_t)stream);
}

extern "C" int fmi_upfirdn2d_nhwc_bf16(const uint16_t* in, const float* kernel, uint16_t* out, int N, int in_h, int in_w, int C, int kh,
                                       int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                       void* stream) {
  if (!in || !kernel || !out || N <= 0 || in_h <= 0 || in_w <= 0 || C <= 0 || kh <= 0 || kw <= 0) return FMI_ERR_BAD_ARG;
  if (up_x != 1 || up_y != 1 || down_x != 1 || down_y != 1 || kh != kw || kh < 2 || kh > 4 || C % 8 != 0 || !al16(in) || !al16(out))
    return FMI_ERR_UNSUPPORTED;  // the bf16 decoder only blurs; resampling stays on the fp32 entry
  const int fh = in_h + pad_y0 + pad_y1 - kh, fw = in_w + pad_x0 + pad_x1 - kw;
  if (fh < 0 || fw < 0) return FMI_ERR_BAD_ARG;
  const int out_h = fh + 1, out_w = fw + 1;
  if (fmi_internal_fir_run_launch(in, kernel, out, N, in_h, in_w, C / 8, out_h, out_w, kh, pad_x0, pad_y0, 1, stream)) return fmi_launch_status();
  const int64_t tv = (int64_t)N * out_h * ((out_w + 1) / 2) * (C / 8);
  if (tv >= (1ll << 31)) return FMI_ERR_UNSUPPORTED;
  const int grid = fmi_bw_grid(tv, 256);
#define FIR_LAUNCH(K_)                                                                                                          \
  hipLaunchKernelGGL((upfirdn2d_nhwc_fir_bf16_kernel<K_, K_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, \
                     kernel, (uint4*)out, in_h, in_w, C / 8, out_h, out_w, pad_x0, pad_y0, (int)tv)
  if (kh == 4) FIR_LAUNCH(4);
  else if (kh == 3) FIR_LAUNCH(3);
  else FIR_LAUNCH(2);
#undef FIR_LAUNCH
  return fmi_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// ToRGB: out[n][p][o] = sum_c x[n][p][c] * w[o][c] * s[n][c] + bias[o] + skip[n][p][o],   o < 3, out / skip / bias fp32.
// A pixel is spread over C8 = C/8 consecutive lanes (C8 <= 64, a power of two); the three partial dot products meet by butterfly.
__global__ void __launch_bounds__(256) torgb_fwd_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ s, const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ out, int64_t P, int C8,
                                                             int64_t rows_per_block) {
  const int RL = 256 / C8, cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y, C = C8 * 8;
  float wm[3][8];
  {
    float sc[8];
    load8f(s + (int64_t)n * C + 8 * cg, sc);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      load8f(w + o * C + 8 * cg, wm[o]);
#pragma unroll
      for (int e = 0; e < 8; ++e) wm[o][e] *= sc[e];
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const uint4* xb = x + (int64_t)n * P * C8;
  for (int64_t rb = r0; rb < r1; rb += RL) {  // every lane runs every iteration (the butterfly needs the whole group)
    const int64_t r = rb + rl;
    float d[3] = {0.f, 0.f, 0.f};
    if (r < r1) {
      float v[8];
      unpack8(xb[r * C8 + cg], v);
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) d[o] = fmaf(v[e], wm[o][e], d[o]);
    }
    for (int off = C8 >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int o = 0; o < 3; ++o) d[o] += __shfl_xor(d[o], off, 64);
    if (cg == 0 && r < r1) {
      const int64_t q = ((int64_t)n * P + r) * 3;
#pragma unroll
      for (int o = 0; o < 3; ++o) out[q + o] = d[o] + (bias ? bias[o] : 0.f) + (skip ? skip[q + o] : 0.f);
    }
  }
}
// backward: gx[n][p][c] = sum_o g[n][p][o] w[o][c] s[n][c];  per block one partial row [3][C] of sum_p g[n][p][o] x[n][p][c] followed
// by the three sums of g (8 floats reserved): ws[(n * blocks + b)][3 C + 8]
__global__ void __launch_bounds__(256) torgb_bwd_bf16_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ s, const float* __restrict__ g,
                                                             uint4* __restrict__ gx, float* __restrict__ ws,
                                                             int64_t P, int C8, int64_t rows_per_block) {
  __shared__ float part[256][8];
  __shared__ float red[4];
  const int RL = 256 / C8, cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int n = blockIdx.y, C = C8 * 8;
  float wm[3][8];
  {
    float sc[8];
    load8f(s + (int64_t)n * C + 8 * cg, sc);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      load8f(w + o * C + 8 * cg, wm[o]);
#pragma unroll
      for (int e = 0; e < 8; ++e) wm[o][e] *= sc[e];
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const uint4* xb = x + (int64_t)n * P * C8;
  uint4* gxb = gx + (int64_t)n * P * C8;
  float acc[3][8];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
  float gb[3] = {0.f, 0.f, 0.f};
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    const int64_t q = ((int64_t)n * P + r) * 3;
    const float g0 = g[q], g1 = g[q + 1], g2 = g[q + 2];
    float v[8], o8[8];
    unpack8(xb[r * C8 + cg], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o8[e] = g0 * wm[0][e] + g1 * wm[1][e] + g2 * wm[2][e];
      acc[0][e] = fmaf(g0, v[e], acc[0][e]);
      acc[1][e] = fmaf(g1, v[e], acc[1][e]);
      acc[2][e] = fmaf(g2, v[e], acc[2][e]);
    }
    gxb[r * C8 + cg] = pack8(o8);
    if (cg == 0) gb[0] += g0, gb[1] += g1, gb[2] += g2;
  }
  float* row = ws + ((int64_t)n * gridDim.x + blockIdx.x) * (3 * C + 8);
#pragma unroll
  for (int o = 0; o < 3; ++o) chunk_reduce_store(acc[o], row + o * C, C8, part);
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float t = block_sum_256(gb[o], red);
    if (threadIdx.x == 0) row[3 * C + o] = t;
  }
}
// gwm[n][3C+8] = per-sample sums of the partial rows (sum_parts_kernel); then
// gw[o][c] = sum_n gwm[n][o][c] s[n][c];   gs[n][c] = sum_o gwm[n][o][c] w[o][c];   gbias[o] = sum_n gwm[n][3C + o]
__global__ void __launch_bounds__(256) torgb_finish_kernel(const float* __restrict__ gwm, const float* __restrict__ w,
                                                           const float* __restrict__ s, float* __restrict__ gw, float* __restrict__ gs,
                                                           float* __restrict__ gbias, int N, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int W = 3 * C + 8;
  if (i < 3 * C) {
    const int c = i % C, o = i / C;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a = fmaf(gwm[(int64_t)n * W + o * C + c], s[(int64_t)n * C + c], a);
    gw[i] = a;
  }
  if (i < N * C) {
    const int c = i % C, n = i / C;
    float a = 0.f;
#pragma unroll
    for (int o = 0; o < 3; ++o) a = fmaf(gwm[(int64_t)n * W + o * C + c], w[o * C + c], a);
    gs[i] = a;
  }
  if (gbias && i < 3) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += gwm[(int64_t)n * W + 3 * C + i];
    gbias[i] = a;
  }
}
static bool torgb_ok(int C) {
  const int c8 = C / 8;
  return C % 8 == 0 && c8 >= 1 && c8 <= 64 && (c8 & (c8 - 1)) == 0;
}
extern "C" int fmi_torgb_fwd_bf16(const uint16_t* x, const float* w, const float* s, const float* bias, const float* skip, float* out,
                                  int N, int64_t P, int C, void* stream) {
  if (!x || !w || !s || !out || N <= 0 || P <= 0 || C <= 0 || N > 65535) return FMI_ERR_BAD_ARG;
  if (!torgb_ok(C) || !al16(x) || !al16(w) || !al16(s)) return FMI_ERR_UNSUPPORTED;
  const int RL = 256 / (C / 8);
  int64_t blocks = ceil_div64(P, 64);
  if (blocks > 512) blocks = 512;
  int64_t rpb = ceil_div64(ceil_div64(P, blocks), RL) * RL;
  blocks = ceil_div64(P, rpb);
  hipLaunchKernelGGL(torgb_fwd_bf16_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, w, s, bias, skip,
                     out, P, C / 8, rpb);
  return fmi_launch_status();
}
/* ws: partials workspace of ws_floats >= 2 * N * (3 C + 8) floats (more = more row blocks); gw [3][C], gs [N][C], gbias [3] (may be
 * NULL) are written, nothing has to be zeroed */
extern "C" int fmi_torgb_bwd_bf16(const uint16_t* x, const float* w, const float* s, const float* g, uint16_t* gx, float* ws,
                                  int64_t ws_floats, float* gw, float* gs, float* gbias, int N, int64_t P, int C, void* stream) {
  if (!x || !w || !s || !g || !gx || !ws || !gw || !gs || N <= 0 || P <= 0 || C <= 0 || N > 65535) return FMI_ERR_BAD_ARG;
  const int64_t W = 3 * (int64_t)C + 8;
  if (ws_floats < 2 * N * W) return FMI_ERR_BAD_ARG;
  if (!torgb_ok(C) || !al16(x) || !al16(gx) || !al16(w) || !al16(s) || !al16(ws)) return FMI_ERR_UNSUPPORTED;
  float* gwm = ws;                 // [N][W] per-sample sums
  float* parts = ws + N * W;       // [N][blocks][W]
  int64_t blocks = parts_for(P, N, W, ws_floats - N * W);
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(torgb_bwd_bf16_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, (const uint4*)x, w, s, g, (uint4*)gx, parts, P,
                     C / 8, rpb);
  launch_sum_parts(parts, gwm, (int)blocks, (int)W, N, st);
  const int tot = (N > 3 ? N : 3) * C;
  hipLaunchKernelGGL(torgb_finish_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, gwm, w, s, gw, gs, gbias, N, C);
  return fmi_launch_status();
}
#endif  // FMI_HOST_EMU
