// Weight preparation for a whole network in ONE launch: spectral normalisation (one power iteration,
// external_function.py:30-41) fused with the re-layout of the effective weight into the two packed forms
// the implicit-GEMM kernels read, its backward, and the multi-tensor Adam update.
// One 256-thread workgroup per weight tensor (<= 2.4 MB each on the hot path; 95 tensors run side by side
// on separate CUs); these kernels stream each weight a few times through one CU and are launch-count, not
// bandwidth, optimisations: they replace ~6 tiny ATen kernels per conv per forward.
#include "common.h"
#include "x6.h"

#define SN_MAX_ROWS 4096
#define ENTRY_CHUNK 32
struct PrepArgs { fmi_weight_entry e[ENTRY_CHUNK]; int it; };  // it: power iteration this launch belongs to (entries with fewer iterations sit it out)
__host__ __device__ inline int sn_iters(const fmi_weight_entry& e) { return e.iters > 1 ? e.iters : 1; }
struct GradArgs { fmi_weight_grad_entry e[ENTRY_CHUNK]; };
struct AdamArgs { fmi_adam_entry e[ENTRY_CHUNK * 2]; };

// Every stage is a 2-D grid: blockIdx.y = weight tensor (<= 32 per launch), blockIdx.x = slice of that tensor
// (surplus slices exit).  Five small, fully parallel launches replace one workgroup streaming a 2.4 MB weight
// four times (1.4 ms per launch in the first version of this file).

// stage 1: v_raw[j] = sum_r W[r][j] u[r]   (thread per column, coalesced along the row; written into v)
__global__ void __launch_bounds__(256) wp_wtu_kernel(const PrepArgs args) {
  __shared__ float su[SN_MAX_ROWS];
  const fmi_weight_entry e = args.e[blockIdx.y];
  if (!e.u || args.it >= sn_iters(e)) return;
  const int width = e.C * e.taps, j0 = blockIdx.x * 256;
  if (j0 >= width) return;
  for (int i = threadIdx.x; i < e.rows; i += 256) su[i] = e.u[i];
  __syncthreads();
  const int j = j0 + threadIdx.x;
  if (j >= width) return;
  float s = 0.f;
  for (int r = 0; r < e.rows; ++r) s += e.w[(int64_t)r * width + j] * su[r];
  e.v[j] = s;
}
// stage 2: v = v_raw / (||v_raw|| + 1e-12)
__global__ void __launch_bounds__(256) wp_vnorm_kernel(const PrepArgs args) {
  __shared__ float red[4];
  const fmi_weight_entry e = args.e[blockIdx.x];
  if (!e.u || args.it >= sn_iters(e)) return;
  const int width = e.C * e.taps;
  float n = 0.f;
  for (int j = threadIdx.x; j < width; j += 256) n += e.v[j] * e.v[j];
  n = sqrtf(block_sum_256(n, red));
  const float d = n + 1e-12f;
  for (int j = threadIdx.x; j < width; j += 256) e.v[j] = e.v[j] / d;
}
// stage 3: t[r] = W[r,:] . v  (one wave per row; written into u, whose old value is no longer needed)
__global__ void __launch_bounds__(256) wp_wv_kernel(const PrepArgs args) {
  const fmi_weight_entry e = args.e[blockIdx.y];
  if (!e.u || args.it >= sn_iters(e)) return;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= e.rows) return;
  const int width = e.C * e.taps;
  float s = 0.f;
  for (int j = lane; j < width; j += 64) s += e.w[(int64_t)r * width + j] * e.v[j];
  s = wave_sum(s);
  if (lane == 0) e.u[r] = s;
}
// stage 4: u = t / (||t|| + 1e-12); sigma = u . t
__global__ void __launch_bounds__(256) wp_unorm_kernel(const PrepArgs args) {
  __shared__ float red[4];
  const fmi_weight_entry e = args.e[blockIdx.x];
  if (!e.u || args.it >= sn_iters(e)) return;
  float n = 0.f;
  for (int i = threadIdx.x; i < e.rows; i += 256) n += e.u[i] * e.u[i];
  n = sqrtf(block_sum_256(n, red));
  const float d = n + 1e-12f;
  float sg = 0.f;
  for (int i = threadIdx.x; i < e.rows; i += 256) {
    const float t = e.u[i], un = t / d;
    e.u[i] = un;
    sg += un * t;
  }
  sg = block_sum_256(sg, red);
  if (threadIdx.x == 0) e.sigma[0] = sg;
}
// stage 5: packed copies of W / sigma: wf[tap][c][row] (row fastest) and wt[tap][row][c] (c fastest)
#define PACK_PER_BLOCK 2048
// The re-layouts (w[row][c][tap] <-> wf[tap][c][row] / wt[tap][row][c]) go through an LDS tile of 32 rows x CC channels x all taps:
// every global access is a run of >= 128 contiguous bytes on both sides.  (The first version gathered 4-byte elements `rows` apart:
// 0.5 TB/s on the 200 M parameters of the IR-SE50 encoder, 4.7 ms per pSp step.)
#define TR_R 32
#define TR_MAX 288  // CC * taps <= 288 floats per row of the tile (32 channels at 9 taps)
__host__ __device__ inline int tr_cc(int taps) {
  int cc = TR_MAX / taps;
  if (cc > 32) cc = 32;
  if (cc >= 8) cc &= ~7;  // whole groups of 8 channels per tile: a 16-byte chunk of the bf16 piece images never straddles two tiles
  return cc < 1 ? 1 : cc;
}
__host__ __device__ inline int64_t tr_tiles(int rows, int C, int taps) {
  const int cc = tr_cc(taps);
  return (int64_t)((rows + TR_R - 1) / TR_R) * ((C + cc - 1) / cc);
}
// One tile of the re-layout.  TAPS_ > 0: the tap count at compile time (9 and 1 are all the networks have), FULL: a whole 32-row x
// tr_cc-channel tile -- every index split below then divides by constants (with run-time divisors the kernel spent its time in integer
// division: 2.2 ms per train_psp step for 24 bytes per weight, a third of what the memory system moves in that time).
template <int TAPS_, bool FULL>
__device__ __forceinline__ void wp_tile(const fmi_weight_entry& e, float (*tile)[TR_MAX + 1], int r0, int c0, int cn_rt, int rn_rt, int taps_rt, int cc_rt) {
  const int rows = e.rows, C = e.C;
  const int taps = TAPS_ > 0 ? TAPS_ : taps_rt;
  const int cc = TAPS_ > 0 ? tr_cc(TAPS_) : cc_rt;
  const int cn = FULL ? cc : cn_rt, rn = FULL ? TR_R : rn_rt;
  const int width = C * taps, run = cn * taps;
  const float sigma = e.u ? e.sigma[0] : 1.f;
  for (int i = threadIdx.x; i < rn * run; i += 256) {  // w: for each row a contiguous run of cn * taps floats
    const int rl = i / run, j = i - rl * run;
    float v = e.w[(int64_t)(r0 + rl) * width + c0 * taps + j];
    if (e.u) v = v / sigma;
    tile[rl][j] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TR_R * cn * taps; i += 256) {  // wf[tap][c][row]: 32 consecutive rows per (tap, c)
    const int rl = i % TR_R, q = i / TR_R, cl = q % cn, tap = q / cn;
    if (rl < rn) e.wf[((int64_t)tap * C + c0 + cl) * rows + r0 + rl] = tile[rl][cl * taps + tap];
  }
  if (e.wt) {
    for (int i = threadIdx.x; i < rn * cn * taps; i += 256) {  // wt[tap][row][c]: cn consecutive channels per (tap, row)
      const int cl = i % cn, q = i / cn, rl = q % rn, tap = q / rn;
      e.wt[((int64_t)tap * rows + r0 + rl) * C + c0 + cl] = tile[rl][cl * taps + tap];
    }
  }
  // bf16 piece images for the bf16x6 products (x6.h, ConvWX3 in gemm_core.h): one 16-byte chunk = 8 consecutive REDUCTION indices of one
  // output column, three images (the exact three-way split of the fp32 value)
  const int64_t pstride = (int64_t)taps * C * rows;
  if (e.wf3 && (C & 7) == 0 && (cc & 7) == 0) {  // wf3[piece][tap][C / 8][rows][8]: reduction = channels
    const int ncg = cn >> 3;
    for (int i = threadIdx.x; i < taps * ncg * rn; i += 256) {
      const int rl = i % rn, q = i / rn, cg = q % ncg, tap = q / ncg;
      uint32_t w[3][4];
#pragma unroll
      for (int h = 0; h < 4; ++h)
        split3_pair(tile[rl][(cg * 8 + 2 * h) * taps + tap], tile[rl][(cg * 8 + 2 * h + 1) * taps + tap], w[0][h], w[1][h], w[2][h]);
      const int64_t chunk = ((int64_t)tap * (C >> 3) + (c0 >> 3) + cg) * rows + r0 + rl;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(e.wf3) + pc * pstride + chunk * 8) = make_uint4(w[pc][0], w[pc][1], w[pc][2], w[pc][3]);
    }
  }
  if (e.wt3 && (rows & 7) == 0) {  // wt3[piece][tap][rows / 8][C][8]: reduction = rows
    const int nrg = rn >> 3;
    for (int i = threadIdx.x; i < taps * nrg * cn; i += 256) {
      const int cl = i % cn, q = i / cn, rg = q % nrg, tap = q / nrg;
      uint32_t w[3][4];
#pragma unroll
      for (int h = 0; h < 4; ++h)
        split3_pair(tile[rg * 8 + 2 * h][cl * taps + tap], tile[rg * 8 + 2 * h + 1][cl * taps + tap], w[0][h], w[1][h], w[2][h]);
      const int64_t chunk = ((int64_t)tap * (rows >> 3) + (r0 >> 3) + rg) * C + c0 + cl;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(e.wt3) + pc * pstride + chunk * 8) = make_uint4(w[pc][0], w[pc][1], w[pc][2], w[pc][3]);
    }
  }
}
__global__ void __launch_bounds__(256) wp_pack_kernel(const PrepArgs args) {
  __shared__ float tile[TR_R][TR_MAX + 1];
  const fmi_weight_entry e = args.e[blockIdx.y];
  const int rows = e.rows, C = e.C, taps = e.taps, width = C * taps;
  if (taps > TR_MAX) {  // no tile fits: element-wise gather
    const int64_t total = (int64_t)rows * width;
    const float sigma = e.u ? e.sigma[0] : 1.f;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
      const int r = (int)(o % rows);
      const int64_t q = o / rows;
      const int c = (int)(q % C), tap = (int)(q / C);
      float v = e.w[(int64_t)r * width + c * taps + tap];
      if (e.u) v = v / sigma;
      e.wf[o] = v;
      if (e.wt) e.wt[((int64_t)tap * rows + r) * C + c] = v;
    }
    return;
  }
  const int cc = tr_cc(taps), ctiles = (C + cc - 1) / cc;
  if ((int64_t)blockIdx.x >= tr_tiles(rows, C, taps)) return;
  const int r0 = (blockIdx.x / ctiles) * TR_R, c0 = (blockIdx.x % ctiles) * cc;
  const int cn = C - c0 < cc ? C - c0 : cc, rn = rows - r0 < TR_R ? rows - r0 : TR_R, run = cn * taps;
  if (taps == 9) {
    if (cn == cc && rn == TR_R) wp_tile<9, true>(e, tile, r0, c0, cn, rn, taps, cc);
    else wp_tile<9, false>(e, tile, r0, c0, cn, rn, taps, cc);
  } else if (taps == 1) {
    if (cn == cc && rn == TR_R) wp_tile<1, true>(e, tile, r0, c0, cn, rn, taps, cc);
    else wp_tile<1, false>(e, tile, r0, c0, cn, rn, taps, cc);
  } else {
    wp_tile<0, false>(e, tile, r0, c0, cn, rn, taps, cc);
  }
}

extern "C" int fmi_weight_prepare_f32(const fmi_weight_entry* entries, int count, void* stream) {
  if (!entries || count <= 0) return FMI_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    const fmi_weight_entry& e = entries[i];
    if (!e.w || !e.wf || e.rows <= 0 || e.C <= 0 || e.taps <= 0) return FMI_ERR_BAD_ARG;
    if (e.u && (!e.v || !e.sigma)) return FMI_ERR_BAD_ARG;
    if (e.u && e.rows > SN_MAX_ROWS) return FMI_ERR_UNSUPPORTED;
    if (e.iters < 0 || e.iters > 64) return FMI_ERR_BAD_ARG;
    // piece images are written by the tiled path only (taps <= TR_MAX, 8-channel groups inside a tile): anything else would leave
    // the caller's buffer uninitialised while fmi_conv2d_* trusted it
    if (e.wf3 && (e.taps > TR_MAX || (tr_cc(e.taps) & 7) || (e.C & 7))) return FMI_ERR_UNSUPPORTED;
    if (e.wt3 && (e.taps > TR_MAX || (e.rows & 7))) return FMI_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += ENTRY_CHUNK) {
    PrepArgs a;
    const int n = count - base < ENTRY_CHUNK ? count - base : ENTRY_CHUNK;
    int max_w = 1, max_r = 1, any_sn = 0;
    int64_t max_t = 1;
    for (int i = 0; i < n; ++i) {
      a.e[i] = entries[base + i];
      const int w = a.e[i].C * a.e[i].taps;
      if (w > max_w) max_w = w;
      if (a.e[i].rows > max_r) max_r = a.e[i].rows;
      if ((int64_t)w * a.e[i].rows > max_t) max_t = (int64_t)w * a.e[i].rows;
      if (a.e[i].u) any_sn = 1;
    }
    int max_it = 0;
    for (int i = 0; i < n; ++i)
      if (a.e[i].u && sn_iters(a.e[i]) > max_it) max_it = sn_iters(a.e[i]);
    for (a.it = 0; a.it < (any_sn ? max_it : 0); ++a.it) {  // external_function.py:36: for _ in range(power_iterations)
      hipLaunchKernelGGL(wp_wtu_kernel, dim3((max_w + 255) / 256, n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(wp_vnorm_kernel, dim3(n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(wp_wv_kernel, dim3((max_r + 3) / 4, n), dim3(256), 0, st, a);
      hipLaunchKernelGGL(wp_unorm_kernel, dim3(n), dim3(256), 0, st, a);
    }
    a.it = 0;
    int64_t max_tiles = 1;
    for (int i = 0; i < n; ++i) {
      const int64_t t = a.e[i].taps > TR_MAX ? 1024 : tr_tiles(a.e[i].rows, a.e[i].C, a.e[i].taps);
      if (t > max_tiles) max_tiles = t;
    }
    hipLaunchKernelGGL(wp_pack_kernel, dim3((unsigned)max_tiles, n), dim3(256), 0, st, a);
  }
  return fmi_launch_status();
}

// dW = dWeff / sigma - (sum dWeff o W) / sigma^2 * u v^T      (u, v read live)
// stage 1: dots[e] += partial sum of dWeff o W ; stage 2: the element-wise formula
__global__ void __launch_bounds__(256) wg_dot_kernel(const GradArgs args, float* __restrict__ dots) {
  __shared__ float red[4];
  const fmi_weight_grad_entry e = args.e[blockIdx.y];
  if (!e.u) return;
  const int rows = e.rows, C = e.C, taps = e.taps, width = C * taps;
  const int64_t total = (int64_t)rows * width;
  if ((int64_t)blockIdx.x * PACK_PER_BLOCK >= total) return;
  float s = 0.f;
  for (int64_t base = (int64_t)blockIdx.x * PACK_PER_BLOCK; base < total; base += (int64_t)gridDim.x * PACK_PER_BLOCK) {  // one chunk per block; ALL chunks in reproducible mode (grid.x = 1)
    for (int64_t o = base + threadIdx.x; o < base + PACK_PER_BLOCK && o < total; o += 256) {  // o indexes dwf: [tap][c][row]
      const int r = (int)(o % rows);
      const int64_t q = o / rows;
      const int c = (int)(q % C), tap = (int)(q / C);
      s += e.dwf[o] * e.w[(int64_t)r * width + c * taps + tap];
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(dots + blockIdx.y, s);
}
// one tile of the gradient's way back (dwf[tap][c][row] -> dw[row][c][tap]), specialised like wp_tile
template <int TAPS_, bool FULL>
__device__ __forceinline__ void wg_tile(const fmi_weight_grad_entry& e, float (*tile)[TR_MAX + 1], int r0, int c0, int cn_rt, int rn_rt, int taps_rt,
                                        float sigma, float coef) {
  const int rows = e.rows, C = e.C;
  const int taps = TAPS_ > 0 ? TAPS_ : taps_rt;
  const int cn = (FULL && TAPS_ > 0) ? tr_cc(TAPS_ > 0 ? TAPS_ : 1) : cn_rt, rn = FULL ? TR_R : rn_rt;
  const int width = C * taps, run = cn * taps;
  for (int i = threadIdx.x; i < TR_R * cn * taps; i += 256) {  // dwf[tap][c][row]: 32 consecutive rows per (tap, c)
    const int rl = i % TR_R, q = i / TR_R, cl = q % cn, tap = q / cn;
    if (rl < rn) tile[rl][cl * taps + tap] = e.dwf[((int64_t)tap * C + c0 + cl) * rows + r0 + rl];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rn * run; i += 256) {  // dw[row][c][tap]: a contiguous run of cn * taps floats per row
    const int rl = i / run, j = i - rl * run;
    float g = tile[rl][j];
    if (e.u) g = g / sigma - coef * e.u[r0 + rl] * e.v[c0 * taps + j];
    e.dw[(int64_t)(r0 + rl) * width + c0 * taps + j] = g;
  }
}
__global__ void __launch_bounds__(256) wg_apply_kernel(const GradArgs args, const float* __restrict__ dots) {
  __shared__ float tile[TR_R][TR_MAX + 1];
  const fmi_weight_grad_entry e = args.e[blockIdx.y];
  const int rows = e.rows, C = e.C, taps = e.taps, width = C * taps;
  float sigma = 1.f, coef = 0.f;
  if (e.u) {
    sigma = e.sigma[0];
    coef = dots[blockIdx.y] / (sigma * sigma);
  }
  if (taps > TR_MAX) {  // no tile fits: element-wise gather
    const int64_t total = (int64_t)rows * width;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {  // o indexes dw: [row][c][tap]
      const int j = (int)(o % width);
      const int r = (int)(o / width);
      const int c = j / taps, tap = j - c * taps;
      float g = e.dwf[((int64_t)tap * C + c) * rows + r];
      if (e.u) g = g / sigma - coef * e.u[r] * e.v[j];
      e.dw[o] = g;
    }
    return;
  }
  const int cc = tr_cc(taps), ctiles = (C + cc - 1) / cc;
  if ((int64_t)blockIdx.x >= tr_tiles(rows, C, taps)) return;
  const int r0 = (blockIdx.x / ctiles) * TR_R, c0 = (blockIdx.x % ctiles) * cc;
  const int cn = C - c0 < cc ? C - c0 : cc, rn = rows - r0 < TR_R ? rows - r0 : TR_R;
  if (taps == 9) {
    if (cn == cc && rn == TR_R) wg_tile<9, true>(e, tile, r0, c0, cn, rn, taps, sigma, coef);
    else wg_tile<9, false>(e, tile, r0, c0, cn, rn, taps, sigma, coef);
  } else if (taps == 1) {
    if (cn == cc && rn == TR_R) wg_tile<1, true>(e, tile, r0, c0, cn, rn, taps, sigma, coef);
    else wg_tile<1, false>(e, tile, r0, c0, cn, rn, taps, sigma, coef);
  } else {
    wg_tile<0, false>(e, tile, r0, c0, cn, rn, taps, sigma, coef);
  }
}
extern "C" int fmi_weight_grad_f32(const fmi_weight_grad_entry* entries, int count, float* scratch_zeroed, void* stream) {
  if (!entries || count <= 0 || !scratch_zeroed) return FMI_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    const fmi_weight_grad_entry& e = entries[i];
    if (!e.w || !e.dwf || !e.dw || e.rows <= 0 || e.C <= 0 || e.taps <= 0) return FMI_ERR_BAD_ARG;
    if (e.u && (!e.v || !e.sigma)) return FMI_ERR_BAD_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += ENTRY_CHUNK) {
    GradArgs a;
    const int n = count - base < ENTRY_CHUNK ? count - base : ENTRY_CHUNK;
    int64_t max_t = 1;
    int any_sn = 0;
    for (int i = 0; i < n; ++i) {
      a.e[i] = entries[base + i];
      const int64_t t = (int64_t)a.e[i].C * a.e[i].taps * a.e[i].rows;
      if (t > max_t) max_t = t;
      if (a.e[i].u) any_sn = 1;
    }
    const dim3 grid(fmi_det() ? 1u : (unsigned)ceil_div64(max_t, PACK_PER_BLOCK), n);  // reproducible mode: one block per tensor
    if (any_sn) hipLaunchKernelGGL(wg_dot_kernel, grid, dim3(256), 0, st, a, scratch_zeroed + base);
    int64_t max_tiles = 1;
    for (int i = 0; i < n; ++i) {
      const int64_t t = a.e[i].taps > TR_MAX ? 1024 : tr_tiles(a.e[i].rows, a.e[i].C, a.e[i].taps);
      if (t > max_tiles) max_tiles = t;
    }
    hipLaunchKernelGGL(wg_apply_kernel, dim3((unsigned)max_tiles, n), dim3(256), 0, st, a, (const float*)(scratch_zeroed + base));
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

// The same update with the step count in DEVICE memory (incremented by the launch sequence itself): nothing about the step is baked
// into the launch arguments, so a training step captured in a HIP graph replays with the right bias corrections.
// guard (may be null): a device scalar, normally the loss of the step; a non-finite value turns the whole step into a no-op (counter
// included) -- the "skip the step if the loss is not finite" of train_psp.py:328-331 as a device-side predicate, so that a HIP-graph
// replay needs no host read
__device__ __forceinline__ bool adam_guard_ok(const float* guard) { return !guard || isfinite(guard[0]); }
__global__ void adam_step_inc_kernel(int* step, const float* guard) {
  if (adam_guard_ok(guard)) step[0] += 1;
}
__global__ void __launch_bounds__(256) adam_dev_kernel(const AdamArgs args, float lr, float beta1, float beta2, float eps, float wd,
                                                       const int* __restrict__ step, const float* __restrict__ guard) {
  const fmi_adam_entry e = args.e[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.x * ADAM_CHUNK;
  if (base >= e.n || !adam_guard_ok(guard)) return;
  const int st = step[0];
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)st));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)st));
  const float step_size = lr / bc1;
  for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < e.n; i += 256) {
    float g = e.g[i];
    const float p = e.p[i];
    if (wd != 0.f) g += wd * p;
    const float m = e.m[i] + (g - e.m[i]) * (1.f - beta1);
    const float v = e.v[i] * beta2 + (1.f - beta2) * (g * g);
    e.m[i] = m;
    e.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    e.p[i] = p - step_size * (m / denom);
  }
}
extern "C" int fmi_adam_step_dev_guarded_f32(const fmi_adam_entry* entries, int count, float lr, float beta1, float beta2, float eps,
                                             float weight_decay, int* step_dev, const float* guard, void* stream);
extern "C" int fmi_adam_step_dev_f32(const fmi_adam_entry* entries, int count, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int* step_dev, void* stream) {
  return fmi_adam_step_dev_guarded_f32(entries, count, lr, beta1, beta2, eps, weight_decay, step_dev, nullptr, stream);
}
extern "C" int fmi_adam_step_dev_guarded_f32(const fmi_adam_entry* entries, int count, float lr, float beta1, float beta2, float eps,
                                             float weight_decay, int* step_dev, const float* guard, void* stream) {
  if (!entries || count <= 0 || !step_dev) return FMI_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, st, step_dev, guard);
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
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)gx, n), dim3(256), 0, st, a, lr, beta1, beta2, eps, weight_decay, (const int*)step_dev, guard);
  }
  return fmi_launch_status();
}

// ---- multi-tensor Ranger = RAdam + Lookahead + gradient centralisation (modules/psp/ranger.py:92-184) ----------------------
// launch 1 (tensors of more than gc_dim dimensions): mean[r] of every gradient row; launch 2: the element-wise update in the
// reference's order -- g -= mean[row]; v = v beta2 + (1 - beta2) g g; m = m beta1 + (1 - beta1) g; weight decay; p -= step_size lr
// m / (sqrt(v) + eps) (rectified) or p -= step_size lr m; every k-th step slow += alpha (p - slow), p = slow
struct RangerArgs { fmi_ranger_entry e[ENTRY_CHUNK]; };
__global__ void __launch_bounds__(256) ranger_rowmean_kernel(const RangerArgs args) {
  __shared__ double red[4];
  const fmi_ranger_entry e = args.e[blockIdx.y];
  if (!e.row_mean) return;
  const int64_t rows = e.n / e.cols;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < e.cols; c += 256) s += (double)e.g[r * e.cols + c];
    s = block_sum_256_d(s, red);
    if (threadIdx.x == 0) e.row_mean[r] = (float)(s / (double)e.cols);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) ranger_kernel(const RangerArgs args, float lr, float beta1, float beta2, float eps, float wd,
                                                     float step_size, int rectified, float alpha, int lookahead) {
  const fmi_ranger_entry e = args.e[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.x * ADAM_CHUNK;
  if (base >= e.n) return;
  for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < e.n; i += 256) {
    float g = e.g[i];
    if (e.row_mean) g -= e.row_mean[i / e.cols];
    const float v = e.v[i] * beta2 + (1.f - beta2) * g * g;
    const float m = e.m[i] * beta1 + (1.f - beta1) * g;
    e.v[i] = v;
    e.m[i] = m;
    float p = e.p[i];
    if (wd != 0.f) p += -wd * lr * p;
    if (rectified)
      p += -step_size * lr * (m / (sqrtf(v) + eps));
    else
      p += -step_size * lr * m;
    if (lookahead) {
      const float s = e.slow[i] + alpha * (p - e.slow[i]);
      e.slow[i] = s;
      p = s;
    }
    e.p[i] = p;
  }
}
extern "C" int fmi_ranger_step_f32(const fmi_ranger_entry* entries, int count, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float step_size, int rectified, float alpha, int lookahead, void* stream) {
  if (!entries || count <= 0) return FMI_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += ENTRY_CHUNK) {
    RangerArgs a;
    const int n = count - base < ENTRY_CHUNK ? count - base : ENTRY_CHUNK;
    int64_t mx = 0, max_rows = 0;
    for (int i = 0; i < n; ++i) {
      a.e[i] = entries[base + i];
      const fmi_ranger_entry& e = a.e[i];
      if (!e.p || !e.g || !e.m || !e.v || !e.slow || e.n <= 0 || e.cols <= 0 || e.n % e.cols) return FMI_ERR_BAD_ARG;
      if (e.n > mx) mx = e.n;
      if (e.row_mean && e.n / e.cols > max_rows) max_rows = e.n / e.cols;
    }
    if (max_rows > 0)
      hipLaunchKernelGGL(ranger_rowmean_kernel, dim3((unsigned)(max_rows < 4096 ? max_rows : 4096), n), dim3(256), 0, st, a);
    const int64_t gx = ceil_div64(mx, ADAM_CHUNK);
    if (gx > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ranger_kernel, dim3((unsigned)gx, n), dim3(256), 0, st, a, lr, beta1, beta2, eps, weight_decay, step_size, rectified,
                       alpha, lookahead);
  }
  return fmi_launch_status();
}
