// Loss reductions: L1 / MSE / LSGAN sums, the Gram-matrix style-loss gradient, and the contextual loss
// (external_function.py:231-274) forward/backward tails that sit around the cosine-similarity GEMM.
#include "common.h"

// ---- sum reductions -----------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float term(float a, float b, float c0) {
  if (KIND == 0) return fabsf(a - b);
  if (KIND == 1) return (a - b) * (a - b);
  return (a - c0) * (a - c0);
}
template <int KIND>
__global__ void __launch_bounds__(256) reduce_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                          float c0, float scale, float* __restrict__ out) {
  __shared__ double red[4];
  float s = 0.f;
  double acc = 0.0;
  int cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    s += term<KIND>(a[i], KIND == 2 ? 0.f : b[i], c0);
    if (++cnt == 64) {  // flush the fp32 run into fp64
      acc += s;
      s = 0.f;
      cnt = 0;
    }
  }
  acc += s;
  acc = block_sum_256_d(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, (float)(acc * (double)scale));
}
extern "C" int fmi_reduce_loss_f32(int kind, const float* a, const float* b, int64_t n, float c0, float scale, float* out,
                                   void* stream) {
  if (!a || !out || n <= 0 || (kind != 2 && !b)) return FMI_ERR_BAD_ARG;
  const int grid = fmi_det() ? 1 : fmi_bw_grid(n, 256 * 16);  // reproducible mode: one block, one contribution
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL((reduce_loss_kernel<0>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, out);
  else if (kind == 1) hipLaunchKernelGGL((reduce_loss_kernel<1>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, out);
  else if (kind == 2) hipLaunchKernelGGL((reduce_loss_kernel<2>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, out);
  else return FMI_ERR_UNSUPPORTED;
  return fmi_launch_status();
}

template <int KIND>
__global__ void __launch_bounds__(256) reduce_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              int64_t n, float c0, float scale,
                                                              const float* __restrict__ gscale, float* __restrict__ ga) {
  const float g = gscale[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - (KIND == 2 ? c0 : b[i]);
    float v;
    if (KIND == 0) v = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    else v = 2.f * d * g;
    ga[i] = v;
  }
}
extern "C" int fmi_reduce_loss_bwd_f32(int kind, const float* a, const float* b, int64_t n, float c0, float scale,
                                       const float* gscale, float* ga, void* stream) {
  if (!a || !ga || !gscale || n <= 0 || (kind != 2 && !b)) return FMI_ERR_BAD_ARG;
  const int grid = fmi_bw_grid(n, 256);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL((reduce_loss_bwd_kernel<0>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, gscale, ga);
  else if (kind == 1) hipLaunchKernelGGL((reduce_loss_bwd_kernel<1>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, gscale, ga);
  else if (kind == 2) hipLaunchKernelGGL((reduce_loss_bwd_kernel<2>), dim3(grid), dim3(256), 0, st, a, b, n, c0, scale, gscale, ga);
  else return FMI_ERR_UNSUPPORTED;
  return fmi_launch_status();
}

// ---- contextual loss ----------------------------------------------------------------------------
// mu[c] += (1/rows) * sum_rows y[row][c]
__global__ void __launch_bounds__(256) col_mean_kernel(const float* __restrict__ y, float* __restrict__ mu, int64_t rows, int C,
                                                       int64_t rows_per_block, float inv_rows) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int cg = 0; cg < C; cg += 64) {
    const int c = cg + tx;
    float s = 0.f;
    if (c < C)
      for (int64_t r = r0 + ty; r < r1; r += 4) s += y[r * C + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) atomicAdd(mu + c, (part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]) * inv_rows);
    __syncthreads();
  }
}
extern "C" int fmi_cx_channel_mean_f32(const float* y, float* mu, int64_t rows, int C, void* stream) {
  if (!y || !mu || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  int64_t blocks = ceil_div64(rows, 64);
  if (blocks > 1024) blocks = 1024;
  if (fmi_det()) blocks = 1;
  const int64_t rpb = ceil_div64(rows, blocks);
  blocks = ceil_div64(rows, rpb);
  hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, mu, rows, C, rpb,
                     1.f / (float)rows);
  return fmi_launch_status();
}

// one wave per row: out = (x - mu) / ||x - mu||
__global__ void __launch_bounds__(256) cx_normalise_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                           float* __restrict__ out, float* __restrict__ inv_norm, int64_t rows,
                                                           int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - mu[c];
    ss += d * d;
  }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  for (int c = lane; c < C; c += 64) out[row * C + c] = (xr[c] - mu[c]) / nrm;
  if (lane == 0) inv_norm[row] = 1.f / nrm;
}
extern "C" int fmi_cx_normalise_f32(const float* x, const float* mu, float* out, float* inv_norm, int64_t rows, int C,
                                    void* stream) {
  if (!x || !mu || !out || !inv_norm || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_normalise_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, mu, out,
                     inv_norm, rows, C);
  return fmi_launch_status();
}
// gx = (g - xn * (xn . g)) * inv_norm
__global__ void __launch_bounds__(256) cx_normalise_bwd_kernel(const float* __restrict__ g, const float* __restrict__ xn,
                                                               const float* __restrict__ inv_norm, float* __restrict__ gx,
                                                               int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += g[row * C + c] * xn[row * C + c];
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int c = lane; c < C; c += 64) gx[row * C + c] = (g[row * C + c] - xn[row * C + c] * dot) * inv;
}
extern "C" int fmi_cx_normalise_bwd_f32(const float* g, const float* xn, const float* inv_norm, float* gx, int64_t rows, int C,
                                        void* stream) {
  if (!g || !xn || !inv_norm || !gx || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_normalise_bwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, xn,
                     inv_norm, gx, rows, C);
  return fmi_launch_status();
}

// one workgroup per row i of cos[n][i][:]: d = 1-cos, dmin, w = exp((1 - d/(dmin+1e-5))/h), cx = w / sum(w)
__global__ void __launch_bounds__(256) cx_rows_kernel(const float* __restrict__ cosm, float* __restrict__ cxij,
                                                      float* __restrict__ dmin, int* __restrict__ argmin,
                                                      float* __restrict__ rowsum, int P, float h) {
  __shared__ float red[4];
  __shared__ int redi[4];
  const int64_t row = blockIdx.x;
  const float* cr = cosm + row * P;
  float m = INFINITY;
  int am = 0x7fffffff;
  for (int j = threadIdx.x; j < P; j += 256) {
    const float d = 1.f - cr[j];
    if (d < m) {
      m = d;
      am = j;
    }
  }
  // block argmin (smallest index among ties)
  float wm = m;
  int wa = am;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(wm, off, 64);
    const int oa = __shfl_xor(wa, off, 64);
    if (om < wm || (om == wm && oa < wa)) {
      wm = om;
      wa = oa;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = wm;
    redi[threadIdx.x >> 6] = wa;
  }
  __syncthreads();
  float gm = red[0];
  int ga = redi[0];
  for (int w = 1; w < 4; ++w)
    if (red[w] < gm || (red[w] == gm && redi[w] < ga)) {
      gm = red[w];
      ga = redi[w];
    }
  __syncthreads();
  const float den = gm + 1e-5f;
  float s = 0.f;
  for (int j = threadIdx.x; j < P; j += 256) s += expf((1.f - (1.f - cr[j]) / den) / h);
  s = block_sum_256(s, red);
  for (int j = threadIdx.x; j < P; j += 256) cxij[row * P + j] = expf((1.f - (1.f - cr[j]) / den) / h) / s;
  if (threadIdx.x == 0) {
    dmin[row] = gm;
    argmin[row] = ga;
    rowsum[row] = s;
  }
}
extern "C" int fmi_cx_rows_f32(const float* cosm, float* cxij, float* dmin, int* argmin, float* rowsum, int N, int P, float h,
                               void* stream) {
  if (!cosm || !cxij || !dmin || !argmin || !rowsum || N <= 0 || P <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_rows_kernel, dim3((unsigned)(N * P)), dim3(256), 0, (hipStream_t)stream, cosm, cxij, dmin, argmin, rowsum,
                     P, h);
  return fmi_launch_status();
}

// colmax[n][j] = max_i cx[n][i][j] (first maximum), one thread per column
__global__ void __launch_bounds__(256) cx_cols_kernel(const float* __restrict__ cxij, float* __restrict__ colmax,
                                                      int* __restrict__ colarg, int P) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (j >= P) return;
  const float* b = cxij + (int64_t)n * P * P + j;
  float m = b[0];
  int a = 0;
  for (int i = 1; i < P; ++i) {
    const float v = b[(int64_t)i * P];
    if (v > m) {
      m = v;
      a = i;
    }
  }
  colmax[(int64_t)n * P + j] = m;
  colarg[(int64_t)n * P + j] = a;
}
extern "C" int fmi_cx_cols_f32(const float* cxij, float* colmax, int* colarg, int N, int P, void* stream) {
  if (!cxij || !colmax || !colarg || N <= 0 || P <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_cols_kernel, dim3((P + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, cxij, colmax, colarg, P);
  return fmi_launch_status();
}

// cx[n] = mean_j colmax ; loss += scale * (-log(cx + 1e-5)) / N
__global__ void __launch_bounds__(256) cx_loss_kernel(const float* __restrict__ colmax, float* __restrict__ cx,
                                                      float* __restrict__ loss, int N, int P, float scale) {
  __shared__ float red[4];
  for (int n = blockIdx.x; n < N; n += gridDim.x) {  // one sample per block; all samples in order in reproducible mode (grid = 1)
    float s = 0.f;
    for (int j = threadIdx.x; j < P; j += 256) s += colmax[(int64_t)n * P + j];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
      const float c = s / (float)P;
      cx[n] = c;
      atomicAdd(loss, scale * (-logf(c + 1e-5f)) / (float)N);
    }
  }
}
extern "C" int fmi_cx_loss_f32(const float* colmax, float* cx, float* loss, int N, int P, float scale, void* stream) {
  if (!colmax || !cx || !loss || N <= 0 || P <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_loss_kernel, dim3(fmi_det() ? 1 : N), dim3(256), 0, (hipStream_t)stream, colmax, cx, loss, N, P, scale);
  return fmi_launch_status();
}

// backward to the cosine matrix, one workgroup per row i (derivation in DESIGN.md "contextual loss")
__global__ void __launch_bounds__(256) cx_bwd_kernel(const float* __restrict__ cxij, const float* __restrict__ dmin,
                                                     const int* __restrict__ argmin, const float* __restrict__ cosm,
                                                     const int* __restrict__ colarg, const float* __restrict__ cx,
                                                     const float* __restrict__ gscale, float* __restrict__ dcos, int N, int P,
                                                     float h, float scale) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int n = (int)(row / P), i = (int)(row - (int64_t)n * P);
  // upstream: dL/dcolmax[n][j] = -scale*g / (N * (cx_n + 1e-5) * P), reaching cx_ij only where i == colarg[n][j]
  const float gcol = -scale * gscale[0] / ((float)N * (cx[n] + 1e-5f) * (float)P);
  const float* cr = cxij + row * P;
  const int* ca = colarg + (int64_t)n * P;
  float t = 0.f;
  for (int j = threadIdx.x; j < P; j += 256)
    if (ca[j] == i) t += gcol * cr[j];
  t = block_sum_256(t, red);  // T_i = sum_j G_ij cx_ij
  const float den = dmin[row] + 1e-5f;
  float macc = 0.f;
  for (int j = threadIdx.x; j < P; j += 256) {
    const float G = (ca[j] == i) ? gcol : 0.f;
    const float e = (G - t) * cr[j];  // = dL/dw_ij * w_ij
    const float d = 1.f - cosm[row * P + j];
    macc += e * d;
    dcos[row * P + j] = e / (h * den);  // -(dL/dd_ij) through w
  }
  macc = block_sum_256(macc, red);
  __syncthreads();
  if (threadIdx.x == 0) {
    // dL/d dmin_i = sum_j e_ij d_ij / (h den^2), routed to the arg-min entry; dcos = -dd
    dcos[row * P + argmin[row]] -= macc / (h * den * den);
  }
}
extern "C" int fmi_cx_bwd_f32(const float* cxij, const float* dmin, const int* argmin, const float* rowsum, const float* cosm,
                              const int* colarg, const float* cx, const float* gscale, float* dcos, int N, int P, float h,
                              float scale, void* stream) {
  (void)rowsum;
  if (!cxij || !dmin || !argmin || !cosm || !colarg || !cx || !gscale || !dcos || N <= 0 || P <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(cx_bwd_kernel, dim3((unsigned)(N * P)), dim3(256), 0, (hipStream_t)stream, cxij, dmin, argmin, cosm, colarg, cx,
                     gscale, dcos, N, P, h, scale);
  return fmi_launch_status();
}

// ---- SSIM (modules/evaluations/ssim.py:18-38): 11x11 gaussian window (outer product of g[ws]), zero padding,
// per (plane, pixel) ssim value summed into out[plane_group] with fp64 block sums.  Metric only (no gradient).
__global__ void __launch_bounds__(256) ssim_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ g, int ws, int H, int W, int planes_per_out,
                                                   float* __restrict__ out, int planes) {
  __shared__ double red[4];
  __shared__ float sg[64];
  if (threadIdx.x < ws) sg[threadIdx.x] = g[threadIdx.x];
  __syncthreads();
  const int half = ws / 2;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {  // one plane per block row; all planes in order in reproducible mode
  const float* pa = a + (int64_t)plane * H * W;
  const float* pb = b + (int64_t)plane * H * W;
  double acc = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int dy = 0; dy < ws; ++dy) {
      const int yy = y + dy - half;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = 0; dx < ws; ++dx) {
        const int xx = x + dx - half;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float wgt = sg[dy] * sg[dx];
        const float va = pa[yy * W + xx], vb = pb[yy * W + xx];
        m1 += wgt * va;
        m2 += wgt * vb;
        s11 += wgt * va * va;
        s22 += wgt * vb * vb;
        s12 += wgt * va * vb;
      }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    acc += (double)(((2.f * m12 + C1) * (2.f * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2)));
  }
  acc = block_sum_256_d(acc, red);
  if (threadIdx.x == 0) atomicAdd(out + plane / planes_per_out, (float)acc);
  __syncthreads();
  }
}
extern "C" int fmi_ssim_f32(const float* img1, const float* img2, const float* window1d, int ws, int planes, int H, int W,
                            int planes_per_out, float* out_zeroed, void* stream) {
  if (!img1 || !img2 || !window1d || !out_zeroed || ws <= 0 || ws > 63 || planes <= 0 || H <= 0 || W <= 0 || planes_per_out <= 0 ||
      planes > 65535)
    return FMI_ERR_BAD_ARG;
  int gx = (H * W + 255) / 256;
  if (gx > 64) gx = 64;
  const dim3 grid = fmi_det() ? dim3(1, 1) : dim3(gx, planes);
  hipLaunchKernelGGL(ssim_kernel, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, window1d, ws, H, W, planes_per_out, out_zeroed, planes);
  return fmi_launch_status();
}

// ---- LPIPS / ArcFace pieces (criteria/lpips/utils.py:6-8 normalize_activation, helpers.py:15-18 l2_norm, lpips.py:30-36) ----
// one wave per row: y = x / (||x|| + eps); inv[row] = 1 / (||x|| + eps)
__global__ void __launch_bounds__(256) l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                                          int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += xr[c] * xr[c];
  ss = wave_sum(ss);
  const float s = 1.f / (sqrtf(ss) + eps);
  for (int c = lane; c < C; c += 64) y[row * C + c] = xr[c] * s;
  if (lane == 0) inv[row] = s;
}
// gx = s (g - y (g . y) (n + eps) / n), n = 1 / s - eps; an all-zero row (n = 0: autograd's 0 / 0 in the reference) gets s g
__global__ void __launch_bounds__(256) l2norm_rows_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                              const float* __restrict__ inv, float* __restrict__ gx, int64_t rows, int C,
                                                              float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += g[row * C + c] * y[row * C + c];
  dot = wave_sum(dot);
  const float s = inv[row];
  const float n = 1.f / s - eps;
  const float k = n > 0.f ? dot * (n + eps) / n : 0.f;
  for (int c = lane; c < C; c += 64) gx[row * C + c] = s * (g[row * C + c] - y[row * C + c] * k);
}
extern "C" int fmi_l2norm_rows_f32(const float* x, float* y, float* inv_norm, int64_t rows, int C, float eps, void* stream) {
  if (!x || !y || !inv_norm || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, rows, C, eps);
  return fmi_launch_status();
}
extern "C" int fmi_l2norm_rows_bwd_f32(const float* g, const float* y, const float* inv_norm, float* gx, int64_t rows, int C, float eps,
                                       void* stream) {
  if (!g || !y || !inv_norm || !gx || rows <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, y, inv_norm, gx, rows, C, eps);
  return fmi_launch_status();
}
// out[0] += scale * sum_p sum_c w[c] (fx[p][c] - fy[p][c])^2: one LPIPS layer = squared difference, 1x1 "lin" convolution to one
// channel, spatial mean and the sum over the batch in ONE pass (lpips.py:33-36); per-workgroup partial, one atomic each
__global__ void __launch_bounds__(256) lpips_layer_kernel(const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ w,
                                                          float* __restrict__ out, int64_t total, int C, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float d = fx[i] - fy[i];
    s += w[i % C] * d * d;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(out, s * scale);
}
__global__ void __launch_bounds__(256) lpips_layer_bwd_kernel(const float* __restrict__ fx, const float* __restrict__ fy,
                                                              const float* __restrict__ w, const float* __restrict__ gout,
                                                              float* __restrict__ gfx, float* __restrict__ gfy, int64_t total, int C,
                                                              float scale) {
  const float k = 2.f * scale * gout[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = k * w[i % C] * (fx[i] - fy[i]);
    if (gfx) gfx[i] = v;
    if (gfy) gfy[i] = -v;
  }
}
extern "C" int fmi_lpips_layer_f32(const float* fx, const float* fy, const float* w, float* out, int64_t pixels, int C, float scale,
                                   void* stream) {
  if (!fx || !fy || !w || !out || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * C;
  int g = fmi_det() ? 1 : fmi_bw_grid(total, 256 * 8);
  hipLaunchKernelGGL(lpips_layer_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, fx, fy, w, out, total, C, scale);
  return fmi_launch_status();
}
extern "C" int fmi_lpips_layer_bwd_f32(const float* fx, const float* fy, const float* w, const float* gout, float* gfx, float* gfy,
                                       int64_t pixels, int C, float scale, void* stream) {
  if (!fx || !fy || !w || !gout || (!gfx && !gfy) || pixels <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = pixels * C;
  hipLaunchKernelGGL(lpips_layer_bwd_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, fx, fy, w, gout, gfx, gfy, total, C, scale);
  return fmi_launch_status();
}

// ---- SSIM / MS-SSIM in the form of the trainers' metric package (pytorch_msssim, absent offline: train_reference_fill.py:207-209,
// train_psp.py:176-178, PICNet_inference.py:130-131): Gaussian window WITHOUT padding ("valid" filtering), per plane the means of the
// ssim map and of its contrast-structure factor cs; between the scales of MS-SSIM a 2 x 2 mean pool with zero padding of odd sizes.
// Every workgroup writes its partial sums (fp64) and a second launch adds them in a fixed order: results are reproducible.
__global__ void __launch_bounds__(256) ssim_valid_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                                         int ws, int H, int W, float C1, float C2, double* __restrict__ part) {
  __shared__ double red[4];
  __shared__ float sg[64];
  if (threadIdx.x < ws) sg[threadIdx.x] = g[threadIdx.x];
  __syncthreads();
  const int plane = blockIdx.y, oh = H - ws + 1, ow = W - ws + 1;
  const float* pa = a + (int64_t)plane * H * W;
  const float* pb = b + (int64_t)plane * H * W;
  double acc_s = 0.0, acc_c = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < oh * ow; i += gridDim.x * 256) {
    const int y = i / ow, x = i - y * ow;
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int dy = 0; dy < ws; ++dy) {
      const float* ra = pa + (int64_t)(y + dy) * W + x;
      const float* rb = pb + (int64_t)(y + dy) * W + x;
      float r1 = 0.f, r2 = 0.f, r11 = 0.f, r22 = 0.f, r12 = 0.f;  // the row first, then the column weight: the separable order
      for (int dx = 0; dx < ws; ++dx) {
        const float wgt = sg[dx], va = ra[dx], vb = rb[dx];
        r1 += wgt * va;
        r2 += wgt * vb;
        r11 += wgt * va * va;
        r22 += wgt * vb * vb;
        r12 += wgt * va * vb;
      }
      const float wy = sg[dy];
      m1 += wy * r1, m2 += wy * r2, s11 += wy * r11, s22 += wy * r22, s12 += wy * r12;
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float cs = (2.f * (s12 - m12) + C2) / ((s11 - m11) + (s22 - m22) + C2);
    acc_c += (double)cs;
    acc_s += (double)(((2.f * m12 + C1) / (m11 + m22 + C1)) * cs);
  }
  acc_s = block_sum_256_d(acc_s, red);
  acc_c = block_sum_256_d(acc_c, red);
  if (threadIdx.x == 0) {
    part[((int64_t)plane * gridDim.x + blockIdx.x) * 2 + 0] = acc_s;
    part[((int64_t)plane * gridDim.x + blockIdx.x) * 2 + 1] = acc_c;
  }
}
__global__ void ssim_valid_sum_kernel(const double* __restrict__ part, int gx, int planes, double inv_count, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * 2) return;
  const int plane = i >> 1, which = i & 1;
  double s = 0.0;
  for (int j = 0; j < gx; ++j) s += part[((int64_t)plane * gx + j) * 2 + which];
  out[i] = (float)(s * inv_count);
}
extern "C" int fmi_ssim_valid_f32(const float* img1, const float* img2, const float* window1d, int ws, int planes, int H, int W, float C1,
                                  float C2, float* out_plane2, double* ws_part, int64_t ws_doubles, void* stream) {
  if (!img1 || !img2 || !window1d || !out_plane2 || !ws_part || ws <= 0 || ws > 63 || planes <= 0 || planes > 65535 || H < ws || W < ws)
    return FMI_ERR_BAD_ARG;
  const int64_t px = (int64_t)(H - ws + 1) * (W - ws + 1);
  int gx = (int)((px + 255) / 256);
  if (gx > 64) gx = 64;
  if (ws_doubles < (int64_t)planes * gx * 2) return FMI_ERR_BAD_ARG;
  hipLaunchKernelGGL(ssim_valid_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)stream, img1, img2, window1d, ws, H, W, C1, C2, ws_part);
  hipLaunchKernelGGL(ssim_valid_sum_kernel, dim3((planes * 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws_part, gx, planes, 1.0 / (double)px,
                     out_plane2);
  return fmi_launch_status();
}
// 2 x 2 mean pool, stride 2, zero padding ph / pw (0 or 1) counted in the divisor (avg_pool2d's count_include_pad default)
__global__ void __launch_bounds__(256) avgpool2_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH, int OW, int ph,
                                                           int pw, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const int64_t r = i / OW;
    const int oy = (int)(r % OH);
    const int64_t plane = r / OH;
    const float* p = x + plane * H * W;
    float s = 0.f;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = 2 * oy - ph + dy, xx = 2 * ox - pw + dx;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) s += p[(int64_t)yy * W + xx];
      }
    y[i] = s * 0.25f;
  }
}
extern "C" int fmi_avgpool2_pad_f32(const float* x, float* y, int planes, int H, int W, int pad_h, int pad_w, void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_h > 1 || pad_w < 0 || pad_w > 1) return FMI_ERR_BAD_ARG;
  const int OH = (H + 2 * pad_h - 2) / 2 + 1, OW = (W + 2 * pad_w - 2) / 2 + 1;
  if (OH <= 0 || OW <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)planes * OH * OW;
  hipLaunchKernelGGL(avgpool2_pad_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, pad_h, pad_w, total);
  return fmi_launch_status();
}
