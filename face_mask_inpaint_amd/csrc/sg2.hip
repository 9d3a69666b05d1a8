// StyleGAN2 decoder helpers on NHWC activations (stylegan2/model.py:187-369): the modulation is applied to the
// ACTIVATIONS (x * s[n,c] before the conv, * demod[n,o] after it), which is algebraically the reference's
// per-sample weight modulation (model.py:244-250) but lets every sample share one packed weight and one GEMM.
// All kernels are bandwidth-class.
#include "common.h"

// y[n][p][c] = x[n][p][c] * s[n][c]
__global__ void __launch_bounds__(256) scale_channels_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             float* __restrict__ y, int64_t P, int C, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t n = i / (P * C);
    y[i] = x[i] * s[n * C + c];
  }
}
__global__ void __launch_bounds__(256) scale_channels_vec_kernel(const float4* __restrict__ x, const float4* __restrict__ s,
                                                                 float4* __restrict__ y, int64_t PC4, int C4, int64_t total4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t n = i / PC4;
    float4 v = x[i];
    const float4 f = s[n * C4 + c4];
    v.x *= f.x, v.y *= f.y, v.z *= f.z, v.w *= f.w;
    y[i] = v;
  }
}
extern "C" int fmi_scale_channels_f32(const float* x, const float* s, float* y, int N, int64_t P, int C, void* stream) {
  if (!x || !s || !y || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)s)) & 15) == 0) {
    const int64_t total4 = (int64_t)N * P * (C / 4);
    hipLaunchKernelGGL(scale_channels_vec_kernel, dim3(fmi_bw_grid(total4, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)s, (float4*)y, P * (C / 4), C / 4, total4);
    return fmi_launch_status();
  }
  const int64_t total = (int64_t)N * P * C;
  hipLaunchKernelGGL(scale_channels_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, s, y, P, C, total);
  return fmi_launch_status();
}
// y = x * s[n][c] + res: the SE gate and the residual add of a bottleneck_IR_SE block (helpers.py:64-72,116-118) in one pass
__global__ void __launch_bounds__(256) scale_channels_add_vec_kernel(const float4* __restrict__ x, const float4* __restrict__ s,
                                                                     const float4* __restrict__ res, float4* __restrict__ y, int64_t PC4,
                                                                     int C4, int64_t total4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t n = i / PC4;
    float4 v = x[i];
    const float4 f = s[n * C4 + c4], r = res[i];
    v.x = v.x * f.x + r.x, v.y = v.y * f.y + r.y, v.z = v.z * f.z + r.z, v.w = v.w * f.w + r.w;
    y[i] = v;
  }
}
extern "C" int fmi_scale_channels_add_f32(const float* x, const float* s, const float* res, float* y, int N, int64_t P, int C, void* stream) {
  if (!x || !s || !res || !y || N <= 0 || P <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)s) | ((uintptr_t)res)) & 15) != 0) return FMI_ERR_UNSUPPORTED;
  const int64_t total4 = (int64_t)N * P * (C / 4);
  hipLaunchKernelGGL(scale_channels_add_vec_kernel, dim3(fmi_bw_grid(total4, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (const float4*)s, (const float4*)res, (float4*)y, P * (C / 4), C / 4, total4);
  return fmi_launch_status();
}
// gs[n][c] = sum_p g[n][p][c] * x[n][p][c].  With a partials workspace (ws_floats >= N*C) every row block stores its sums as one row
// and a second launch adds the rows (gs WRITTEN); without one the blocks add onto the caller-zeroed gs with fp32 atomics, which
// serialise per address (52 us per launch on the SE gates of the pSp encoder).
__global__ void __launch_bounds__(256) scale_channels_gs_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                float* __restrict__ gs, int64_t P, int C, int64_t rows_per_block,
                                                                int to_parts) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const float* gb = g + (int64_t)n * P * C;
  const float* xb = x + (int64_t)n * P * C;
  float* dst = to_parts ? gs + ((int64_t)n * gridDim.x + blockIdx.x) * C : gs + (int64_t)n * C;
  for (int cg = 0; cg < C; cg += 64) {
    const int c = cg + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
      int64_t r = r0 + ty;
      for (; r + 4 < r1; r += 8) {  // two independent pairs of loads in flight
        const float a0 = gb[r * C + c], b0 = xb[r * C + c], a1 = gb[(r + 4) * C + c], b1 = xb[(r + 4) * C + c];
        s0 = fmaf(a0, b0, s0), s1 = fmaf(a1, b1, s1);
      }
      for (; r < r1; r += 4) s0 = fmaf(gb[r * C + c], xb[r * C + c], s0);
    }
    part[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < C) {
      const float t = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
      if (to_parts) dst[c] = t;
      else atomicAdd(dst + c, t);
    }
    __syncthreads();
  }
}
// out[n][c] = sum_{b < nparts} ws[(n * nparts + b) * C + c]
__global__ void __launch_bounds__(256) scale_channels_gs_finish_kernel(const float* __restrict__ ws, float* __restrict__ out, int nparts, int C,
                                                                       int total) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int n = t / C, c = t - n * C;
  const float* p = ws + (int64_t)n * nparts * C + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int q = 0;
  for (; q + 3 < nparts; q += 4) a0 += p[(int64_t)q * C], a1 += p[(int64_t)(q + 1) * C], a2 += p[(int64_t)(q + 2) * C], a3 += p[(int64_t)(q + 3) * C];
  for (; q < nparts; ++q) a0 += p[(int64_t)q * C];
  out[t] = (a0 + a1) + (a2 + a3);
}
extern "C" int fmi_scale_channels_gs_f32(const float* g, const float* x, float* gs, float* ws, int64_t ws_floats, int N, int64_t P, int C,
                                         void* stream) {
  if (!g || !x || !gs || N <= 0 || P <= 0 || C <= 0 || N > 65535) return FMI_ERR_BAD_ARG;
  const bool parts = ws && ws_floats >= (int64_t)N * C;
  int64_t blocks = ceil_div64(P, 128);
  int64_t cap = parts ? ws_floats / ((int64_t)N * C) : 1024;
  const int64_t want = ceil_div64(2048, N);
  if (parts && cap > want) cap = want;
  if (blocks > cap) blocks = cap;
  if (fmi_det() && !parts) blocks = 1;  // reproducible mode without a partials workspace: one block per sample
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipLaunchKernelGGL(scale_channels_gs_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, g, x, parts ? ws : gs, P, C, rpb,
                     parts ? 1 : 0);
  if (parts)
    hipLaunchKernelGGL(scale_channels_gs_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, gs, (int)blocks, C,
                       N * C);
  return fmi_launch_status();
}

// Both gradients of y = x * s[n][c] in ONE pass over g and x: gx = g * s (written) and gs[n][c] = sum_p g * x (row-block partials in ws,
// then the adding launch above) -- the SE gate of every IR-SE block, the modulation / demodulation products of the fp32 decoder.  Two
// launches read g twice, and the gs kernel used 4-byte loads (2.3 TB/s).  A thread keeps one 4-channel group; C % 4 == 0, C / 4 <= 256.
__global__ void __launch_bounds__(256) scale_channels_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ x,
                                                                 const float* __restrict__ s, float4* __restrict__ gx, float* __restrict__ ws,
                                                                 int64_t P, int C4, int64_t rows_per_block) {
  __shared__ float part[256][4];
  const int RL = 256 / C4, cg = threadIdx.x % C4, rl = threadIdx.x / C4;
  const int n = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const float4* gb = g + (int64_t)n * P * C4;
  const float4* xb = x + (int64_t)n * P * C4;
  float4* ob = gx + (int64_t)n * P * C4;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    const float4 sv = *reinterpret_cast<const float4*>(s + ((int64_t)n * C4 + cg) * 4);
    auto one = [&](int64_t r, const float4 gv, const float4 xv) {
      a[0] = fmaf(gv.x, xv.x, a[0]), a[1] = fmaf(gv.y, xv.y, a[1]), a[2] = fmaf(gv.z, xv.z, a[2]), a[3] = fmaf(gv.w, xv.w, a[3]);
      ob[r * C4 + cg] = make_float4(gv.x * sv.x, gv.y * sv.y, gv.z * sv.z, gv.w * sv.w);
    };
    int64_t r = r0 + rl;
    for (; r + RL < r1; r += 2 * RL) {  // two independent pairs of 16-byte loads in flight
      const float4 g0 = gb[r * C4 + cg], x0 = xb[r * C4 + cg], g1 = gb[(r + RL) * C4 + cg], x1 = xb[(r + RL) * C4 + cg];
      one(r, g0, x0);
      one(r + RL, g1, x1);
    }
    for (; r < r1; r += RL) one(r, gb[r * C4 + cg], xb[r * C4 + cg]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[threadIdx.x][e] = a[e];
  __syncthreads();
  if ((int)threadIdx.x < C4) {
    float t[4] = {part[threadIdx.x][0], part[threadIdx.x][1], part[threadIdx.x][2], part[threadIdx.x][3]};
    for (int l = 1; l < RL; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += part[l * C4 + cg][e];
    *reinterpret_cast<float4*>(ws + ((int64_t)n * gridDim.x + blockIdx.x) * C4 * 4 + cg * 4) = make_float4(t[0], t[1], t[2], t[3]);
  }
}
/* gx = g * s[n][c] and gs[n][c] = sum_p g * x in one pass (the adjoint of fmi_scale_channels_f32 / the product part of
 * fmi_scale_channels_add_f32).  ws: >= N * C floats of scratch (more = more row blocks). */
extern "C" int fmi_scale_channels_bwd_f32(const float* g, const float* x, const float* s, float* gx, float* gs, float* ws, int64_t ws_floats, int N,
                                          int64_t P, int C, void* stream) {
  if (!g || !x || !s || !gx || !gs || !ws || N <= 0 || P <= 0 || C <= 0 || N > 65535 || ws_floats < (int64_t)N * C) return FMI_ERR_BAD_ARG;
  if (C % 4 != 0 || C / 4 > 256 || (((uintptr_t)g | (uintptr_t)x | (uintptr_t)s | (uintptr_t)gx | (uintptr_t)ws) & 15)) return FMI_ERR_UNSUPPORTED;
  int64_t blocks = ceil_div64(P, 64);
  int64_t cap = ws_floats / ((int64_t)N * C);
  const int64_t want = ceil_div64(4096, N);
  if (cap > want) cap = want;
  if (blocks > cap) blocks = cap;
  const int64_t rpb = ceil_div64(P, blocks);
  blocks = ceil_div64(P, rpb);
  hipLaunchKernelGGL(scale_channels_bwd_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (const float4*)x, s,
                     (float4*)gx, ws, P, C / 4, rpb);
  hipLaunchKernelGGL(scale_channels_gs_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, gs, (int)blocks, C, N * C);
  return fmi_launch_status();
}

// out[r] = sum_k x[r][k]^2  and its gradient gx[r][k] = 2 x[r][k] g[r]     (W^2 summed over the taps)
__global__ void __launch_bounds__(256) sqsum_last_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int k) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += x[r * k + j] * x[r * k + j];
    out[r] = s;
  }
}
extern "C" int fmi_sqsum_last_f32(const float* x, float* out, int64_t rows, int k, void* stream) {
  if (!x || !out || rows <= 0 || k <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(sqsum_last_kernel, dim3(fmi_bw_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, out, rows, k);
  return fmi_launch_status();
}
__global__ void __launch_bounds__(256) sqsum_last_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             float* __restrict__ gx, int64_t total, int k) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) gx[i] = 2.f * x[i] * g[i / k];
}
extern "C" int fmi_sqsum_last_bwd_f32(const float* x, const float* g, float* gx, int64_t rows, int k, void* stream) {
  if (!x || !g || !gx || rows <= 0 || k <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = rows * k;
  hipLaunchKernelGGL(sqsum_last_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, g, gx, total, k);
  return fmi_launch_status();
}

// backward of y = lrelu(x + bias[c] + nw*noise[p]) * scale:  gx = g * scale * (y > 0 ? 1 : alpha);  gnw += sum gx*noise
__global__ void __launch_bounds__(256) noise_bias_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                 const float* __restrict__ noise, float* __restrict__ gx,
                                                                 float* __restrict__ gnw, int64_t total, int C, float alpha,
                                                                 float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = g[i] * scale * (y[i] > 0.f ? 1.f : alpha);
    gx[i] = v;
    if (noise) acc += v * noise[i / C];
  }
  if (noise && gnw) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(gnw, acc);
  }
}
__global__ void __launch_bounds__(256) noise_bias_act_bwd_vec_kernel(const float4* __restrict__ g, const float4* __restrict__ y,
                                                                     const float* __restrict__ noise, float4* __restrict__ gx,
                                                                     float* __restrict__ gnw, int64_t total4, int C4, float alpha,
                                                                     float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    float4 v = g[i];
    const float4 o = y[i];
    v.x *= scale * (o.x > 0.f ? 1.f : alpha);
    v.y *= scale * (o.y > 0.f ? 1.f : alpha);
    v.z *= scale * (o.z > 0.f ? 1.f : alpha);
    v.w *= scale * (o.w > 0.f ? 1.f : alpha);
    gx[i] = v;
    if (noise) acc += ((v.x + v.y) + (v.z + v.w)) * noise[i / C4];
  }
  if (noise && gnw) {
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(gnw, acc);
  }
}
extern "C" int fmi_noise_bias_act_bwd_f32(const float* g, const float* y, const float* noise, float* gx, float* gnw,
                                          int64_t pixels, int C, float alpha, float scale, void* stream) {
  if (!g || !y || !gx || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 == 0 && ((((uintptr_t)g) | ((uintptr_t)y) | ((uintptr_t)gx)) & 15) == 0) {
    const int64_t total4 = pixels * (C / 4);
    int grid = (fmi_det() && noise && gnw) ? 1 : fmi_bw_grid(total4, 256 * 4);  // reproducible mode: one block sums the noise-weight gradient
    if (grid > 1024) grid = 1024;  // one atomic per block on a single address
    hipLaunchKernelGGL(noise_bias_act_bwd_vec_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (const float4*)y,
                       noise, (float4*)gx, gnw, total4, C / 4, alpha, scale);
    return fmi_launch_status();
  }
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(noise_bias_act_bwd_kernel, dim3((fmi_det() && noise && gnw) ? 1 : fmi_bw_grid(total, 256 * 4)), dim3(256), 0, (hipStream_t)stream, g, y, noise,
                     gx, gnw, total, C, alpha, scale);
  return fmi_launch_status();
}

// upfirdn2d on NHWC tensors (minor dimension = channels): same semantics as fmi_upfirdn2d_f32 per (n, c) plane
__device__ __forceinline__ int fdiv_i(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}
__global__ void __launch_bounds__(256) upfirdn2d_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ kernel,
                                                             float* __restrict__ out, int in_h, int in_w, int C, int out_h,
                                                             int out_w, int kh, int kw, int up_x, int up_y, int down_x,
                                                             int down_y, int pad_x0, int pad_y0, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % out_w);
    r /= out_w;
    const int oy = (int)(r % out_h);
    const int n = (int)(r / out_h);
    const int uy0 = oy * down_y - pad_y0, ux0 = ox * down_x - pad_x0;
    int ky0 = (-uy0) % up_y, kx0 = (-ux0) % up_x;
    if (ky0 < 0) ky0 += up_y;
    if (kx0 < 0) kx0 += up_x;
    float acc = 0.f;
    for (int ky = ky0; ky < kh; ky += up_y) {
      const int iy = (uy0 + ky) / up_y;
      if ((unsigned)iy >= (unsigned)in_h) continue;
      for (int kx = kx0; kx < kw; kx += up_x) {
        const int ix = (ux0 + kx) / up_x;
        if ((unsigned)ix >= (unsigned)in_w) continue;
        acc += in[(((int64_t)n * in_h + iy) * in_w + ix) * C + c] * kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)];
      }
    }
    out[i] = acc;
  }
}
// Vector form for the decoder's Blur (op/upfirdn2d.py:142-147 with up = down = 1: StyledConv up-convolutions, model.py:52-68, and
// their gradients): C % 4 == 0, one thread per (2 adjacent output pixels, 4 channels).  The KH x (KW+1) input window of the
// pair is read once as 16-byte loads (consecutive threads = consecutive channel chunks: coalesced), the taps are wave-uniform
// scalars.  Algorithmic traffic = in + out bytes; the window overlap between neighbouring threads is served by L1/L2.
template <int KH, int KW>
__global__ void __launch_bounds__(256) upfirdn2d_nhwc_fir_kernel(const float* __restrict__ in, const float* __restrict__ kernel,
                                                                 float* __restrict__ out, int in_h, int in_w, int C4, int out_h,
                                                                 int out_w, int pad_x0, int pad_y0, int total) {
  float kf[KH][KW];  // flipped taps: out[oy][ox] = sum_{a,b} in[oy - pad_y0 + a][ox - pad_x0 + b] * kernel[KH-1-a][KW-1-b]
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int b = 0; b < KW; ++b) kf[a][b] = kernel[(KH - 1 - a) * KW + (KW - 1 - b)];
  const int pw = (out_w + 1) >> 1;  // pixel pairs per row
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c4 = i % C4;
    int r = i / C4;
    const int px = r % pw;
    r /= pw;
    const int oy = r % out_h, n = r / out_h;
    const int ox = 2 * px;
    const int iy0 = oy - pad_y0, ix0 = ox - pad_x0;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const float4* base = reinterpret_cast<const float4*>(in) + (int64_t)n * in_h * in_w * C4 + c4;
#pragma unroll
    for (int a = 0; a < KH; ++a) {
      const int iy = iy0 + a;
      if ((unsigned)iy >= (unsigned)in_h) continue;
      const float4* row = base + (int64_t)iy * in_w * C4;
#pragma unroll
      for (int b = 0; b <= KW; ++b) {
        const int ix = ix0 + b;
        if ((unsigned)ix >= (unsigned)in_w) continue;
        const float4 v = row[(int64_t)ix * C4];
        if (b < KW) {
          const float k0 = kf[a][b];
          a0.x = fmaf(v.x, k0, a0.x), a0.y = fmaf(v.y, k0, a0.y), a0.z = fmaf(v.z, k0, a0.z), a0.w = fmaf(v.w, k0, a0.w);
        }
        if (b > 0) {
          const float k1 = kf[a][b - 1];
          a1.x = fmaf(v.x, k1, a1.x), a1.y = fmaf(v.y, k1, a1.y), a1.z = fmaf(v.z, k1, a1.z), a1.w = fmaf(v.w, k1, a1.w);
        }
      }
    }
    float4* o = reinterpret_cast<float4*>(out) + ((int64_t)(n * out_h + oy) * out_w + ox) * C4 + c4;
    o[0] = a0;
    if (ox + 1 < out_w) o[C4] = a1;
  }
}

// column-run FIR shared with the bf16 path (sg2_bf16.hip); returns 0 when it does not apply
extern "C" int fmi_internal_fir_run_launch(const void* in, const float* kernel, void* out, int N, int in_h, int in_w, int CV, int out_h,
                                           int out_w, int k, int pad_x0, int pad_y0, int bf16, void* stream);
extern "C" int fmi_upfirdn2d_nhwc_f32(const float* in, const float* kernel, float* out, int N, int in_h, int in_w, int C,
                                      int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                      int pad_y0, int pad_y1, void* stream) {
  if (!in || !kernel || !out || N <= 0 || in_h <= 0 || in_w <= 0 || C <= 0 || kh <= 0 || kw <= 0) return FMI_ERR_BAD_ARG;
  if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return FMI_ERR_BAD_ARG;
  const int fh = in_h * up_y + pad_y0 + pad_y1 - kh, fw = in_w * up_x + pad_x0 + pad_x1 - kw;
  if (fh < 0 || fw < 0) return FMI_ERR_BAD_ARG;
  const int out_h = fh / down_y + 1, out_w = fw / down_x + 1;
  const int64_t total = (int64_t)N * out_h * out_w * C;
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && C % 4 == 0 && kh == kw && (kh == 4 || kh == 3 || kh == 2) &&
      (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    if (fmi_internal_fir_run_launch(in, kernel, out, N, in_h, in_w, C / 4, out_h, out_w, kh, pad_x0, pad_y0, 0, stream)) return fmi_launch_status();
    const int64_t tv = (int64_t)N * out_h * ((out_w + 1) / 2) * (C / 4);
    if (tv < (1ll << 31) && (int64_t)N * in_h * in_w * C < (1ll << 40)) {
      const int grid = fmi_bw_grid(tv, 256);
#define FIR_LAUNCH(K_)                                                                                                              \
  hipLaunchKernelGGL((upfirdn2d_nhwc_fir_kernel<K_, K_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, kernel, out, in_h, in_w, \
                     C / 4, out_h, out_w, pad_x0, pad_y0, (int)tv)
      if (kh == 4) FIR_LAUNCH(4);
      else if (kh == 3) FIR_LAUNCH(3);
      else FIR_LAUNCH(2);
#undef FIR_LAUNCH
      return fmi_launch_status();
    }
  }
  hipLaunchKernelGGL(upfirdn2d_nhwc_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, kernel, out, in_h,
                     in_w, C, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, total);
  return fmi_launch_status();
}

// ---- pSp encoder helpers (modules/psp/encoders/helpers.py): PReLU(C), MaxPool2d(1, stride) ------------------------
// y = x > 0 ? x : a[c] * x
__global__ void __launch_bounds__(256) prelu_kernel(const float* __restrict__ x, const float* __restrict__ a, float* __restrict__ y,
                                                    int64_t total, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = v > 0.f ? v : a[i % C] * v;
  }
}
__global__ void __launch_bounds__(256) prelu_vec_kernel(const float4* __restrict__ x, const float4* __restrict__ a, float4* __restrict__ y,
                                                        int64_t total4, int C4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    float4 v = x[i];
    const float4 s = a[i % C4];
    v.x = v.x > 0.f ? v.x : s.x * v.x, v.y = v.y > 0.f ? v.y : s.y * v.y, v.z = v.z > 0.f ? v.z : s.z * v.z, v.w = v.w > 0.f ? v.w : s.w * v.w;
    y[i] = v;
  }
}
extern "C" int fmi_prelu_f32(const float* x, const float* a, float* y, int64_t rows, int C, void* stream) {
  if (!x || !a || !y || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)a)) & 15) == 0) {
    const int64_t total4 = rows * (C / 4);
    hipLaunchKernelGGL(prelu_vec_kernel, dim3(fmi_bw_grid(total4, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)a, (float4*)y, total4, C / 4);
    return fmi_launch_status();
  }
  const int64_t total = rows * C;
  hipLaunchKernelGGL(prelu_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, a, y, total, C);
  return fmi_launch_status();
}
// gx = g * (x > 0 ? 1 : a[c]);  ga[c] += sum_rows g * x * (x <= 0)   (caller zeroes ga)
__global__ void __launch_bounds__(256) prelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ a, float* __restrict__ gx, float* __restrict__ ga,
                                                        int64_t rows, int C, int64_t rows_per_block) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int cg = 0; cg < C; cg += 64) {
    const int c = cg + tx;
    float s = 0.f;
    if (c < C) {
      const float ac = a[c];
      for (int64_t r = r0 + ty; r < r1; r += 4) {
        const float xv = x[r * C + c], gv = g[r * C + c];
        gx[r * C + c] = xv > 0.f ? gv : ac * gv;
        if (xv <= 0.f) s += gv * xv;
      }
    }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) atomicAdd(ga + c, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
    __syncthreads();
  }
}
// 16-byte form (C/4 a power of two <= 256): 256 threads = (256 / C4) rows x C4 four-channel chunks per pass, two row passes in flight;
// the scalar kernel walked 64-channel stripes one after the other with a workgroup barrier each (47 us floor per launch)
__global__ void __launch_bounds__(256) prelu_bwd_vec_kernel(const float4* __restrict__ g, const float4* __restrict__ x,
                                                            const float4* __restrict__ a, float4* __restrict__ gx, float* __restrict__ ga,
                                                            int64_t rows, int C4, int64_t rows_per_block, int to_parts) {
  __shared__ float4 part[256];
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const float4 ac = a[cg];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  auto one = [&](int64_t r, const float4 xv, const float4 gv) {
    float4 o;
    o.x = xv.x > 0.f ? gv.x : ac.x * gv.x, o.y = xv.y > 0.f ? gv.y : ac.y * gv.y;
    o.z = xv.z > 0.f ? gv.z : ac.z * gv.z, o.w = xv.w > 0.f ? gv.w : ac.w * gv.w;
    gx[r * C4 + cg] = o;
    if (xv.x <= 0.f) s.x += gv.x * xv.x;
    if (xv.y <= 0.f) s.y += gv.y * xv.y;
    if (xv.z <= 0.f) s.z += gv.z * xv.z;
    if (xv.w <= 0.f) s.w += gv.w * xv.w;
  };
  int64_t r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {
    const float4 x0 = x[r * C4 + cg], g0 = g[r * C4 + cg], x1 = x[(r + RL) * C4 + cg], g1 = g[(r + RL) * C4 + cg];
    one(r, x0, g0);
    one(r + RL, x1, g1);
  }
  for (; r < r1; r += RL) one(r, x[r * C4 + cg], g[r * C4 + cg]);
  part[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < C4) {
    float4 t = part[threadIdx.x];
    for (int l = 1; l < RL; ++l) {
      const float4 v = part[l * C4 + threadIdx.x];
      t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
    }
    if (to_parts) {  // ga = partials workspace: row blockIdx.x
      reinterpret_cast<float4*>(ga)[(int64_t)blockIdx.x * C4 + threadIdx.x] = t;
    } else {
      atomicAdd(ga + 4 * threadIdx.x + 0, t.x);
      atomicAdd(ga + 4 * threadIdx.x + 1, t.y);
      atomicAdd(ga + 4 * threadIdx.x + 2, t.z);
      atomicAdd(ga + 4 * threadIdx.x + 3, t.w);
    }
  }
}
// out[i] = sum_{p < nparts} ws[p * width + i]: 64 columns x 16 row lanes per workgroup, 8 loads in flight per thread
__global__ void __launch_bounds__(1024) sum_parts_f32_kernel(const float* __restrict__ ws, float* __restrict__ out, int nparts, int width) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < width) {
    int q = ty;
    for (; q + 7 * 16 < nparts; q += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(int64_t)(q + 16 * u) * width + i];
    }
    for (; q < nparts; q += 16) a[0] += ws[(int64_t)q * width + i];
  }
  part[ty][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (ty == 0 && i < width) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += part[l][tx];
    out[i] = t;
  }
}
extern "C" int fmi_prelu_bwd_f32(const float* g, const float* x, const float* a, float* gx, float* ga, float* ws, int64_t ws_floats,
                                 int64_t rows, int C, void* stream) {
  if (!g || !x || !a || !gx || !ga || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int C4 = C / 4;
  if (C % 4 == 0 && C4 <= 256 && (C4 & (C4 - 1)) == 0 && ((((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)gx) | ((uintptr_t)a)) & 15) == 0) {
    const bool parts = ws && ws_floats >= C && (((uintptr_t)ws) & 15) == 0;
    int64_t blocks = ceil_div64(rows, 64);
    const int64_t cap = parts ? (ws_floats / C < 1024 ? ws_floats / C : 1024) : 256;  // atomics on C addresses serialise: few blocks
    if (blocks > cap) blocks = cap;
    if (fmi_det() && !parts) blocks = 1;  // reproducible mode without a partials workspace
    const int64_t rpb = ceil_div64(rows, blocks);
    blocks = ceil_div64(rows, rpb);
    hipLaunchKernelGGL(prelu_bwd_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)g, (const float4*)x,
                       (const float4*)a, (float4*)gx, parts ? ws : ga, rows, C4, rpb, parts ? 1 : 0);
    if (parts) hipLaunchKernelGGL(sum_parts_f32_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, ws, ga, (int)blocks, C);
    return fmi_launch_status();
  }
  int64_t blocks = ceil_div64(rows, 128);
  if (blocks > 2048) blocks = 2048;
  if (fmi_det()) blocks = 1;  // reproducible mode
  const int64_t rpb = ceil_div64(rows, blocks);
  blocks = ceil_div64(rows, rpb);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, x, a, gx, ga, rows, C, rpb);
  return fmi_launch_status();
}
// y[n][oy][ox][c] = x[n][oy*s][ox*s][c]  (MaxPool2d(kernel 1, stride s)); backward scatters into a zero-filled gx
__global__ void __launch_bounds__(256) subsample_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                        int OH, int OW, int s, int64_t total, int backward) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    if (!backward) {  // i indexes y
      const int ox = (int)(r % OW);
      r /= OW;
      const int oy = (int)(r % OH);
      const int64_t n = r / OH;
      y[i] = x[((n * H + (int64_t)oy * s) * W + (int64_t)ox * s) * C + c];
    } else {          // i indexes gx (= y argument), x argument = gy
      const int xx = (int)(r % W);
      r /= W;
      const int yy = (int)(r % H);
      const int64_t n = r / H;
      const bool hit = (yy % s == 0) && (xx % s == 0) && (yy / s < OH) && (xx / s < OW);
      y[i] = hit ? x[((n * OH + yy / s) * OW + xx / s) * C + c] : 0.f;
    }
  }
}
extern "C" int fmi_subsample_f32(const float* x, float* y, int N, int H, int W, int C, int stride, int backward, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0) return FMI_ERR_BAD_ARG;
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const int64_t total = backward ? (int64_t)N * H * W * C : (int64_t)N * OH * OW * C;
  hipLaunchKernelGGL(subsample_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, OH, OW, stride,
                     total, backward);
  return fmi_launch_status();
}
