// Weight preparation for a whole network in ONE launch: spectral normalisation (one power iteration,
// external_function.py:30-41) fused with the re-layout of the effective weight into the two packed forms
// the implicit-GEMM kernels read, its backward, and the multi-tensor Adam update.
// One 256-thread workgroup per weight tensor (<= 2.4 MB each on the hot path; 95 tensors run side by side
// on separate CUs); these kernels stream each weight a few times through one CU and are launch-count, not
// bandwidth, optimisations: they replace ~6 tiny ATen kernels per conv per forward.
#include "common.h"

#define SN_MAX_WIDTH 8192  // floats of v kept in LDS
#define SN_MAX_ROWS 1024

#define ENTRY_CHUNK 32
struct PrepArgs { fmi_weight_entry e[ENTRY_CHUNK]; };
struct GradArgs { fmi_weight_grad_entry e[ENTRY_CHUNK]; };
struct AdamArgs { fmi_adam_entry e[ENTRY_CHUNK * 2]; };

__global__ void __launch_bounds__(256) weight_prepare_kernel(const PrepArgs args) {
  const fmi_weight_entry* entries = args.e;
  __shared__ float sv[SN_MAX_WIDTH];
  __shared__ float su[SN_MAX_ROWS];
  __shared__ float red[4];
  const fmi_weight_entry e = entries[blockIdx.x];
  const int rows = e.rows, C = e.C, taps = e.taps, width = C * taps;
  const float* __restrict__ W = e.w;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float sigma = 1.f;
  if (e.u) {
    for (int i = tid; i < rows; i += 256) su[i] = e.u[i];
    __syncthreads();
    // v = normalize(W^T u): thread per column, coalesced along the row
    float nv = 0.f;
    for (int j = tid; j < width; j += 256) {
      float s = 0.f;
      for (int i = 0; i < rows; ++i) s += W[(int64_t)i * width + j] * su[i];
      sv[j] = s;
      nv += s * s;
    }
    nv = sqrtf(block_sum_256(nv, red));
    const float iv = nv + 1e-12f;
    for (int j = tid; j < width; j += 256) {
      const float v = sv[j] / iv;
      sv[j] = v;
      e.v[j] = v;
    }
    __syncthreads();
    // t = W v : one wave per row, lanes stride the row
    float nu = 0.f;
    for (int i = wid; i < rows; i += 4) {
      float s = 0.f;
      for (int j = lane; j < width; j += 64) s += W[(int64_t)i * width + j] * sv[j];
      s = wave_sum(s);
      if (lane == 0) su[i] = s;
      nu += (lane == 0) ? s * s : 0.f;
    }
    nu = sqrtf(block_sum_256(nu, red));  // also orders the su[] writes
    const float iu = nu + 1e-12f;
    float sg = 0.f;
    for (int i = tid; i < rows; i += 256) {
      const float t = su[i], un = t / iu;
      e.u[i] = un;
      sg += un * t;  // sigma = u . (W v)
    }
    sigma = block_sum_256(sg, red);
    if (tid == 0 && e.sigma) e.sigma[0] = sigma;
  }
  // packed copies of W / sigma: wf[tap][c][row] (row fastest) and wt[tap][row][c] (c fastest)
  const int64_t total = (int64_t)rows * width;
  for (int64_t o = tid; o < total; o += 256) {
    const int r = (int)(o % rows);
    const int64_t q = o / rows;
    const int c = (int)(q % C), tap = (int)(q / C);
    float v = W[(int64_t)r * width + c * taps + tap];
    if (e.u) v = v / sigma;
    e.wf[o] = v;
  }
  if (e.wt) {
    for (int64_t o = tid; o < total; o += 256) {
      const int c = (int)(o % C);
      const int64_t q = o / C;
      const int r = (int)(q % rows), tap = (int)(q / rows);
      float v = W[(int64_t)r * width + c * taps + tap];
      if (e.u) v = v / sigma;
      e.wt[o] = v;
    }
  }
}

extern "C" int fmi_weight_prepare_f32(const fmi_weight_entry* entries, int count, void* stream) {
  if (!entries || count <= 0) return FMI_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    const fmi_weight_entry& e = entries[i];
    if (!e.w || !e.wf || e.rows <= 0 || e.C <= 0 || e.taps <= 0) return FMI_ERR_BAD_ARG;
    if (e.u && (!e.v || !e.sigma)) return FMI_ERR_BAD_ARG;
    if (e.u && (e.rows > SN_MAX_ROWS || e.C * e.taps > SN_MAX_WIDTH)) return FMI_ERR_UNSUPPORTED;
  }
  for (int base = 0; base < count; base += ENTRY_CHUNK) {
    PrepArgs a;
    const int n = count - base < ENTRY_CHUNK ? count - base : ENTRY_CHUNK;
    for (int i = 0; i < n; ++i) a.e[i] = entries[base + i];
    hipLaunchKernelGGL(weight_prepare_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
  }
  return fmi_launch_status();
}

// dW = dWeff / sigma - (sum dWeff o W) / sigma^2 * u v^T      (u, v read live)
__global__ void __launch_bounds__(256) weight_grad_kernel(const GradArgs args) {
  const fmi_weight_grad_entry* entries = args.e;
  __shared__ float red[4];
  const fmi_weight_grad_entry e = entries[blockIdx.x];
  const int rows = e.rows, C = e.C, taps = e.taps, width = C * taps;
  const int tid = threadIdx.x;
  const int64_t total = (int64_t)rows * width;
  float sigma = 1.f, coef = 0.f;
  if (e.u) {
    sigma = e.sigma[0];
    float s = 0.f;
    for (int64_t o = tid; o < total; o += 256) {  // o indexes dwf: [tap][c][row]
      const int r = (int)(o % rows);
      const int64_t q = o / rows;
      const int c = (int)(q % C), tap = (int)(q / C);
      s += e.dwf[o] * e.w[(int64_t)r * width + c * taps + tap];
    }
    s = block_sum_256(s, red);
    coef = s / (sigma * sigma);
  }
  for (int64_t o = tid; o < total; o += 256) {  // o indexes dw: [row][c][tap]
    const int j = (int)(o % width);
    const int r = (int)(o / width);
    const int c = j / taps, tap = j - c * taps;
    float g = e.dwf[((int64_t)tap * C + c) * rows + r];
    if (e.u) g = g / sigma - coef * e.u[r] * e.v[j];
    e.dw[o] = g;
  }
}
extern "C" int fmi_weight_grad_f32(const fmi_weight_grad_entry* entries, int count, void* stream) {
  if (!entries || count <= 0) return FMI_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    const fmi_weight_grad_entry& e = entries[i];
    if (!e.w || !e.dwf || !e.dw || e.rows <= 0 || e.C <= 0 || e.taps <= 0) return FMI_ERR_BAD_ARG;
    if (e.u && (!e.v || !e.sigma)) return FMI_ERR_BAD_ARG;
  }
  for (int base = 0; base < count; base += ENTRY_CHUNK) {
    GradArgs a;
    const int n = count - base < ENTRY_CHUNK ? count - base : ENTRY_CHUNK;
    for (int i = 0; i < n; ++i) a.e[i] = entries[base + i];
    hipLaunchKernelGGL(weight_grad_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
  }
  return fmi_launch_status();
}

// ---- multi-tensor Adam (torch.optim.Adam single-tensor arithmetic order) --------------------------
#define ADAM_CHUNK 4096
__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs args, float lr, float beta1,
                                                   float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
  const fmi_adam_entry e = args.e[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.x * ADAM_CHUNK;
  if (base >= e.n) return;
  const float step_size = lr / bc1;
  for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < e.n; i += 256) {
    float g = e.g[i];
    const float p = e.p[i];
    if (wd != 0.f) g += wd * p;
    const float m = e.m[i] + (g - e.m[i]) * (1.f - beta1);  // lerp form used by torch
    const float v = e.v[i] * beta2 + (1.f - beta2) * (g * g);
    e.m[i] = m;
    e.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    e.p[i] = p - step_size * (m / denom);
  }
}
extern "C" int fmi_adam_step_f32(const fmi_adam_entry* entries, int count, int64_t max_n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, void* stream) {
  if (!entries || count <= 0 || max_n <= 0 || step <= 0) return FMI_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  for (int base = 0; base < count; base += ENTRY_CHUNK * 2) {
    AdamArgs a;
    const int n = count - base < ENTRY_CHUNK * 2 ? count - base : ENTRY_CHUNK * 2;
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) {
      a.e[i] = entries[base + i];
      if (!a.e[i].p || !a.e[i].g || !a.e[i].m || !a.e[i].v || a.e[i].n <= 0) return FMI_ERR_BAD_ARG;
      if (a.e[i].n > mx) mx = a.e[i].n;
    }
    const int64_t gx = ceil_div64(mx, ADAM_CHUNK);
    if (gx > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx, n), dim3(256), 0, (hipStream_t)stream, a, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
  }
  return fmi_launch_status();
}
