// Bandwidth-class element-wise kernels: activations, mask handling, example-guided blend, VAE sampling,
// and the reference's fused_bias_act native op.  All are grid-stride, float4-vectorised where the views
// are 16-byte aligned, and sized to fill 256 CUs (common.h: fmi_bw_grid).  Roofline: HBM bytes in + out.
#include "common.h"

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int OP>
__device__ __forceinline__ float ew_op(float a, float b, float p0) {
  if (OP == FMI_EW_LRELU) return a > 0.f ? a : a * p0;
  if (OP == FMI_EW_LRELU_BWD) return b > 0.f ? a : a * p0;
  if (OP == FMI_EW_TANH_BWD) return a * (1.f - b * b);
  if (OP == FMI_EW_ADD) return a + b;
  if (OP == FMI_EW_SCALE) return a * p0;
  if (OP == FMI_EW_AXPY) return p0 * a + b;
  if (OP == FMI_EW_MUL) return a * b;
  if (OP == FMI_EW_RELU_BWD_OUT) return b > 0.f ? a : 0.f;
  if (OP == FMI_EW_SOFTPLUS) return a > 20.f ? a : log1pf(expf(a));
  if (OP == FMI_EW_SOFTPLUS_BWD) return b > 20.f ? a : a / (1.f + expf(-b));
  if (OP == FMI_EW_SUB) return a - b;
  if (OP == FMI_EW_RSQRT) return 1.f / sqrtf(a + p0);
  if (OP == FMI_EW_RSQRT_BWD) return -0.5f * a * b * b * b;
  if (OP == FMI_EW_SIGMOID) return 1.f / (1.f + expf(-a));
  if (OP == FMI_EW_SIGMOID_BWD) return a * b * (1.f - b);
  return 0.f;
}
template <int OP>
constexpr bool ew_binary() {
  return !(OP == FMI_EW_LRELU || OP == FMI_EW_SCALE || OP == FMI_EW_SOFTPLUS || OP == FMI_EW_RSQRT || OP == FMI_EW_SIGMOID);
}

template <int OP, bool VEC>
__global__ void __launch_bounds__(256) eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, int64_t n, float p0) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (VEC) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 va = reinterpret_cast<const float4*>(a)[i];
      float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ew_binary<OP>()) vb = reinterpret_cast<const float4*>(b)[i];
      float4 r;
      r.x = ew_op<OP>(va.x, vb.x, p0);
      r.y = ew_op<OP>(va.y, vb.y, p0);
      r.z = ew_op<OP>(va.z, vb.z, p0);
      r.w = ew_op<OP>(va.w, vb.w, p0);
      reinterpret_cast<float4*>(y)[i] = r;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      y[i] = ew_op<OP>(a[i], ew_binary<OP>() ? b[i] : 0.f, p0);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      y[i] = ew_op<OP>(a[i], ew_binary<OP>() ? b[i] : 0.f, p0);
  }
}

template <int OP>
static int launch_ew(const float* a, const float* b, float* y, int64_t n, float p0, hipStream_t st) {
  if (ew_binary<OP>() && !b) return FMI_ERR_BAD_ARG;
  const bool vec = al16(a) && al16(y) && (!ew_binary<OP>() || al16(b));
  const int grid = fmi_bw_grid(vec ? (n + 3) / 4 : n, 256);
  if (vec) hipLaunchKernelGGL((eltwise_kernel<OP, true>), dim3(grid), dim3(256), 0, st, a, b, y, n, p0);
  else hipLaunchKernelGGL((eltwise_kernel<OP, false>), dim3(grid), dim3(256), 0, st, a, b, y, n, p0);
  return fmi_launch_status();
}

extern "C" int fmi_eltwise_f32(int op, const float* a, const float* b, float* y, int64_t n, float p0, void* stream) {
  if (!a || !y || n <= 0) return FMI_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (op) {
    case FMI_EW_LRELU: return launch_ew<FMI_EW_LRELU>(a, b, y, n, p0, st);
    case FMI_EW_LRELU_BWD: return launch_ew<FMI_EW_LRELU_BWD>(a, b, y, n, p0, st);
    case FMI_EW_TANH_BWD: return launch_ew<FMI_EW_TANH_BWD>(a, b, y, n, p0, st);
    case FMI_EW_ADD: return launch_ew<FMI_EW_ADD>(a, b, y, n, p0, st);
    case FMI_EW_SCALE: return launch_ew<FMI_EW_SCALE>(a, b, y, n, p0, st);
    case FMI_EW_AXPY: return launch_ew<FMI_EW_AXPY>(a, b, y, n, p0, st);
    case FMI_EW_MUL: return launch_ew<FMI_EW_MUL>(a, b, y, n, p0, st);
    case FMI_EW_RELU_BWD_OUT: return launch_ew<FMI_EW_RELU_BWD_OUT>(a, b, y, n, p0, st);
    case FMI_EW_SOFTPLUS: return launch_ew<FMI_EW_SOFTPLUS>(a, b, y, n, p0, st);
    case FMI_EW_SOFTPLUS_BWD: return launch_ew<FMI_EW_SOFTPLUS_BWD>(a, b, y, n, p0, st);
    case FMI_EW_SUB: return launch_ew<FMI_EW_SUB>(a, b, y, n, p0, st);
    case FMI_EW_RSQRT: return launch_ew<FMI_EW_RSQRT>(a, b, y, n, p0, st);
    case FMI_EW_RSQRT_BWD: return launch_ew<FMI_EW_RSQRT_BWD>(a, b, y, n, p0, st);
    case FMI_EW_SIGMOID: return launch_ew<FMI_EW_SIGMOID>(a, b, y, n, p0, st);
    case FMI_EW_SIGMOID_BWD: return launch_ew<FMI_EW_SIGMOID_BWD>(a, b, y, n, p0, st);
    default: return FMI_ERR_UNSUPPORTED;
  }
}

__global__ void __launch_bounds__(256) axpy_dev_kernel(const float* __restrict__ a, const float* __restrict__ s,
                                                       const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  const float sc = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = sc * a[i] + (b ? b[i] : 0.f);
}
extern "C" int fmi_axpy_dev_f32(const float* a, const float* s, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !s || !y || n <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(axpy_dev_kernel, dim3(fmi_bw_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, s, b, y, n);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float scale, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)a[i] * b[i];
  s = block_sum_256_d(s, red);
  if (threadIdx.x == 0) atomicAdd(out, (float)(s * scale));
}
extern "C" int fmi_dot_f32(const float* a, const float* b, int64_t n, float scale, float* out, void* stream) {
  if (!a || !b || !out || n <= 0) return FMI_ERR_BAD_ARG;
  int grid = fmi_det() ? 1 : fmi_bw_grid(n, 256 * 8);
  hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, n, scale, out);
  return fmi_launch_status();
}

// ---- masks -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mask_binarise_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = m[i] > 0 ? 1.f : 0.f;
}
extern "C" int fmi_mask_binarise_i64(const int64_t* mask, float* out, int64_t n, void* stream) {
  if (!mask || !out || n <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(mask_binarise_kernel, dim3(fmi_bw_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, mask, out, n);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                       float* __restrict__ y, int64_t total, int C, int invert) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float mv = m[i / C];
    y[i] = x[i] * (invert ? 1.f - mv : mv);
  }
}
extern "C" int fmi_mask_mul_f32(const float* x, const float* m, float* y, int64_t pixels, int C, int invert, void* stream) {
  if (!x || !m || !y || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, m, y, total, C, invert);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) guide_blend_kernel(const float* __restrict__ ra, const float* __restrict__ rf,
                                                          const float* __restrict__ m, float* __restrict__ out,
                                                          int64_t total, int C, int ocs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const float mv = m[p];
    out[p * ocs + c] = (1.f - mv) * ra[i] + mv * rf[i];
  }
}
extern "C" int fmi_guide_blend_f32(const float* ref_att, const float* ref, const float* m, float* out, int64_t pixels,
                                   int C, int out_cstride, void* stream) {
  if (!ref_att || !ref || !m || !out || pixels <= 0 || C <= 0 || out_cstride < C) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(guide_blend_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, ref_att, ref, m,
                     out, total, C, out_cstride);
  return fmi_launch_status();
}
__global__ void __launch_bounds__(256) guide_blend_bwd_kernel(const float* __restrict__ g, const float* __restrict__ m,
                                                              float* __restrict__ dra, float* __restrict__ drf,
                                                              int64_t total, int C, int gcs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const float mv = m[p], gv = g[p * gcs + c];
    dra[i] = (1.f - mv) * gv;
    drf[i] = mv * gv;
  }
}
extern "C" int fmi_guide_blend_bwd_f32(const float* g, const float* m, float* d_ref_att, float* d_ref, int64_t pixels,
                                       int C, int g_cstride, void* stream) {
  if (!g || !m || !d_ref_att || !d_ref || pixels <= 0 || C <= 0 || g_cstride < C) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(guide_blend_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, g, m,
                     d_ref_att, d_ref, total, C, g_cstride);
  return fmi_launch_status();
}

// ---- VAE re-parameterised sample ------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float a) { return a > 20.f ? a : log1pf(expf(a)); }
__device__ __forceinline__ float sigmoid_sp(float a) { return a > 20.f ? 1.f : 1.f / (1.f + expf(-a)); }

__global__ void __launch_bounds__(256) vae_sample_kernel(const float* __restrict__ os, const float* __restrict__ orf,
                                                         const float* __restrict__ eq, const float* __restrict__ epp,
                                                         float* __restrict__ z, int64_t total, int Z) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / Z;
    const int c = (int)(i - p * Z);
    const int64_t o = p * 2 * Z + c;
    z[o] = os[o] + softplus_f(os[o + Z]) * eq[i];
    z[o + Z] = orf[o] + softplus_f(orf[o + Z]) * epp[i];
  }
}
extern "C" int fmi_vae_sample_f32(const float* o_src, const float* o_ref, const float* eps_q, const float* eps_p, float* z,
                                  int64_t pixels, int Z, void* stream) {
  if (!o_src || !o_ref || !eps_q || !eps_p || !z || pixels <= 0 || Z <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * Z;
  hipLaunchKernelGGL(vae_sample_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, o_src, o_ref,
                     eps_q, eps_p, z, total, Z);
  return fmi_launch_status();
}
__global__ void __launch_bounds__(256) vae_sample_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ os,
                                                             const float* __restrict__ orf, const float* __restrict__ eq,
                                                             const float* __restrict__ epp, float* __restrict__ gs,
                                                             float* __restrict__ gr, int64_t total, int Z) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / Z;
    const int c = (int)(i - p * Z);
    const int64_t o = p * 2 * Z + c;
    const float gq = gz[o], gp = gz[o + Z];
    gs[o] = gq;
    gs[o + Z] = gq * eq[i] * sigmoid_sp(os[o + Z]);
    gr[o] = gp;
    gr[o + Z] = gp * epp[i] * sigmoid_sp(orf[o + Z]);
  }
}
extern "C" int fmi_vae_sample_bwd_f32(const float* gz, const float* o_src, const float* o_ref, const float* eps_q,
                                      const float* eps_p, float* g_src, float* g_ref, int64_t pixels, int Z, void* stream) {
  if (!gz || !o_src || !o_ref || !eps_q || !eps_p || !g_src || !g_ref || pixels <= 0 || Z <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * Z;
  hipLaunchKernelGGL(vae_sample_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gz, o_src,
                     o_ref, eps_q, eps_p, g_src, g_ref, total, Z);
  return fmi_launch_status();
}

// ---- the reference's fused_bias_act (NCHW contiguous; op/fused_bias_act_kernel.cu:36-47 semantics) ----
__device__ __forceinline__ float fba_ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float fba_ld(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }
__device__ __forceinline__ void fba_st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void fba_st(uint16_t* p, int64_t i, float v) {  // bf16, round to nearest even
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
  else u += 0x7fffu + ((u >> 16) & 1u);
  p[i] = (uint16_t)(u >> 16);
}
template <class T>
__global__ void __launch_bounds__(256) fused_bias_act_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                             const T* __restrict__ ref, T* __restrict__ out,
                                                             int64_t n, int step_b, int size_b, int act, int grad,
                                                             float alpha, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = fba_ld(x, i);
    if (bias) v += fba_ld(bias, (i / step_b) % size_b);
    const float r = ref ? fba_ld(ref, i) : 0.f;
    float y;
    if (act == 3) {
      if (grad == 0) y = v > 0.f ? v : v * alpha;
      else if (grad == 1) y = r > 0.f ? v : v * alpha;
      else y = 0.f;
    } else {
      y = grad == 2 ? 0.f : v;
    }
    fba_st(out, i, y * scale);
  }
}
// 16-byte vector form: V consecutive elements share a bias entry when step_b % V == 0
template <class T, int V>
__global__ void __launch_bounds__(256) fused_bias_act_vec_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                                 const T* __restrict__ ref, T* __restrict__ out, int64_t nv, int step_b,
                                                                 int size_b, int act, int grad, float alpha, float scale) {
  struct alignas(16) Pack {
    T e[V];
  };
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const Pack px = reinterpret_cast<const Pack*>(x)[i];
    Pack pr = px, po;
    if (ref) pr = reinterpret_cast<const Pack*>(ref)[i];
    const float b = bias ? fba_ld(bias, ((i * V) / step_b) % size_b) : 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float v = fba_ld(px.e, j) + b;
      const float r = ref ? fba_ld(pr.e, j) : 0.f;
      float y;
      if (act == 3) {
        if (grad == 0) y = v > 0.f ? v : v * alpha;
        else if (grad == 1) y = r > 0.f ? v : v * alpha;
        else y = 0.f;
      } else {
        y = grad == 2 ? 0.f : v;
      }
      fba_st(po.e, j, y * scale);
    }
    reinterpret_cast<Pack*>(out)[i] = po;
  }
}

template <class T>
static int fused_bias_act_launch(const T* x, const T* bias, const T* ref, T* out, int64_t n, int step_b, int size_b, int act, int grad,
                                 float alpha, float scale, void* stream) {
  if (!x || !out || n <= 0) return FMI_ERR_BAD_ARG;
  if (bias && (step_b <= 0 || size_b <= 0)) return FMI_ERR_BAD_ARG;
  if ((act != 1 && act != 3) || grad < 0 || grad > 2) return FMI_ERR_UNSUPPORTED;
  constexpr int V = 16 / (int)sizeof(T);
  const bool al = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)(ref ? ref : x)) & 15) == 0);
  if (al && n % V == 0 && (!bias || step_b % V == 0)) {
    hipLaunchKernelGGL((fused_bias_act_vec_kernel<T, V>), dim3(fmi_bw_grid(n / V, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, ref, out,
                       n / V, step_b, size_b, act, grad, alpha, scale);
    return fmi_launch_status();
  }
  hipLaunchKernelGGL((fused_bias_act_kernel<T>), dim3(fmi_bw_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, ref, out, n,
                     step_b, size_b, act, grad, alpha, scale);
  return fmi_launch_status();
}
extern "C" int fmi_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* out, int64_t n, int step_b,
                                      int size_b, int act, int grad, float alpha, float scale, void* stream) {
  return fused_bias_act_launch<float>(x, bias, ref, out, n, step_b, size_b, act, grad, alpha, scale, stream);
}
extern "C" int fmi_fused_bias_act_bf16(const uint16_t* x, const uint16_t* bias, const uint16_t* ref, uint16_t* out, int64_t n, int step_b,
                                       int size_b, int act, int grad, float alpha, float scale, void* stream) {
  return fused_bias_act_launch<uint16_t>(x, bias, ref, out, n, step_b, size_b, act, grad, alpha, scale, stream);
}

// NHWC variant for StyledConv (stylegan2/model.py:340-346): y = lrelu(x + bias[c] + nw*noise[p]) * scale
__global__ void __launch_bounds__(256) noise_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ noise, const float* __restrict__ nw,
                                                             float* __restrict__ y, int64_t total, int C, float alpha,
                                                             float scale) {
  const float w = (noise && nw) ? nw[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    float v = x[i];
    if (bias) v += bias[c];
    if (noise && nw) v += w * noise[p];
    y[i] = (v > 0.f ? v : v * alpha) * scale;
  }
}
// 16-byte form (C % 4 == 0): the scalar kernel moved 4 bytes per lane and instruction and reached 2.9 TB/s
__global__ void __launch_bounds__(256) noise_bias_act_vec_kernel(const float4* __restrict__ x, const float* __restrict__ bias,
                                                                 const float* __restrict__ noise, const float* __restrict__ nw,
                                                                 float4* __restrict__ y, int64_t total4, int C4, float alpha, float scale) {
  const float w = (noise && nw) ? nw[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C4;
    const int c4 = (int)(i - p * C4);
    float4 v = x[i];
    const float nz = (noise && nw) ? w * noise[p] : 0.f;
    float4 b = make_float4(nz, nz, nz, nz);
    if (bias) {
      const float4 bb = reinterpret_cast<const float4*>(bias)[c4];
      b.x += bb.x, b.y += bb.y, b.z += bb.z, b.w += bb.w;
    }
    v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    v.x = (v.x > 0.f ? v.x : v.x * alpha) * scale;
    v.y = (v.y > 0.f ? v.y : v.y * alpha) * scale;
    v.z = (v.z > 0.f ? v.z : v.z * alpha) * scale;
    v.w = (v.w > 0.f ? v.w : v.w * alpha) * scale;
    y[i] = v;
  }
}
extern "C" int fmi_noise_bias_act_f32(const float* x, const float* bias, const float* noise, const float* nw, float* y,
                                      int64_t pixels, int C, float alpha, float scale, void* stream) {
  if (!x || !y || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  if (C % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)bias)) & 15) == 0) {
    const int64_t total4 = pixels * (C / 4);
    hipLaunchKernelGGL(noise_bias_act_vec_kernel, dim3(fmi_bw_grid(total4, 256 * 2)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       bias, noise, nw, (float4*)y, total4, C / 4, alpha, scale);
    return fmi_launch_status();
  }
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(noise_bias_act_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, noise,
                     nw, y, total, C, alpha, scale);
  return fmi_launch_status();
}

// dbias[c] += sum over (n, hw) of g[n][c][hw]   (NCHW; grad_bias of FusedLeakyReLU, op/fused_act.py:29-36)
__device__ __forceinline__ float bg_load(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float bg_load(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }
template <typename T>
__global__ void __launch_bounds__(256) bias_grad_nchw_kernel(const T* __restrict__ g, int N, int C, int64_t HW,
                                                             float* __restrict__ dbias) {
  __shared__ float red[4];
  // block x = (sample, channel); reproducible mode launches C blocks that walk the samples in order (gridDim.x == C, one block in y)
  const int c = blockIdx.x % C;
  for (int n = blockIdx.x / C; n < N; n += gridDim.x / C) {
    const T* p = g + ((int64_t)n * C + c) * HW;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.y * 256) s += bg_load(p, i);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(dbias + c, s);
  }
}
template <typename T>
static int bias_grad_nchw_launch(const T* g, int N, int C, int64_t HW, float* dbias, void* stream) {
  if (!g || !dbias || N <= 0 || C <= 0 || HW <= 0 || (int64_t)N * C > 0x7fffffffLL) return FMI_ERR_BAD_ARG;
  int gy = (int)((HW + 256 * 16 - 1) / (256 * 16));
  if (gy < 1) gy = 1;
  if (gy > 64) gy = 64;
  const dim3 grid = fmi_det() ? dim3(C, 1) : dim3(N * C, gy);
  hipLaunchKernelGGL(bias_grad_nchw_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, g, N, C, HW, dbias);
  return fmi_launch_status();
}
extern "C" int fmi_bias_grad_nchw_f32(const float* g, int N, int C, int64_t HW, float* dbias, void* stream) {
  return bias_grad_nchw_launch(g, N, C, HW, dbias, stream);
}
// bf16 cotangent (fused_leaky_relu on a bf16 tensor), fp32 sums
extern "C" int fmi_bias_grad_nchw_bf16(const void* g, int N, int C, int64_t HW, float* dbias, void* stream) {
  return bias_grad_nchw_launch((const uint16_t*)g, N, C, HW, dbias, stream);
}

// dst[r][dst_c0 + j] = src[r][src_c0 + j], j < c: channel slices and concatenations of NHWC maps (model.py:106 return_zq,
// unet_parts.py:70 torch.cat, example_guided_att.py:37) without a detour through a temporary
__global__ void __launch_bounds__(256) copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows,
                                                            int sstride, int sc0, int dstride, int dc0, int c, int vec) {
  const int cv = c / vec;
  const int64_t total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cv;
    const int j = (int)(i % cv) * vec;
    if (vec == 4)
      *reinterpret_cast<float4*>(dst + r * dstride + dc0 + j) = *reinterpret_cast<const float4*>(src + r * sstride + sc0 + j);
    else
      dst[r * dstride + dc0 + j] = src[r * sstride + sc0 + j];
  }
}
extern "C" int fmi_copy_channels_f32(const float* src, float* dst, int64_t rows, int src_stride, int src_c0, int dst_stride, int dst_c0,
                                     int c, void* stream) {
  if (!src || !dst || rows <= 0 || c <= 0 || src_c0 < 0 || dst_c0 < 0 || src_c0 + c > src_stride || dst_c0 + c > dst_stride) return FMI_ERR_BAD_ARG;
  const bool v4 = !((src_stride | src_c0 | dst_stride | dst_c0 | c) & 3) && !(((uintptr_t)src | (uintptr_t)dst) & 15);
  const int vec = v4 ? 4 : 1;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(fmi_bw_grid(rows * (c / vec), 256)), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                     src_stride, src_c0, dst_stride, dst_c0, c, vec);
  return fmi_launch_status();
}
