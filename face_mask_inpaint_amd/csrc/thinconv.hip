// 3x3 stride-1 convolution with a THIN output (K <= 4 channels): the generator's Output block
// (base_function.py:367-398: LeakyReLU -> ReflectionPad2d(1) -> conv3x3(32 -> 3) -> tanh) at 1024x1024.
// As a GEMM it has N = 3 columns: the matrix-core kernel runs it at 5 TFLOP/s while the tensor traffic
// (1.07 GB of fp32 activations per batch of 8) needs ~0.2 ms of HBM time.  These kernels are bandwidth kernels:
//
//   lane layout: G = C/4 lanes per pixel, lane j of a group owns channels 4j..4j+3 (one 16-byte load per tap, a group
//   reads a pixel's C floats as one contiguous run); 64/G pixels per wave.  The 9 x 4 x K weights of a lane's
//   channel chunk live in registers for the whole kernel.
//   forward : 9 loads + 36 K FMAs per lane and pixel, xor-shuffle reduction over the G lanes, bias/residual/tanh.
//   dgrad   : dx[q][c] = sum_t sum_k w[t][c][k] dy[.][k]; with reflect padding the fold of the padded gradient is fused:
//             a pixel next to the border also collects the taps that the reflection mapped onto it.
//   (forward and dgrad run as the 4x4x1-MFMA row-segment forms further down; the scalar per-pixel adjoint serves the border pass)
//   wgrad   : per-lane accumulators [9][4][K] over a strided range of pixels, reduced over the wave's pixel groups by
//             shuffles, over the workgroup's waves through LDS, over workgroups by fp32 atomics; dbias rides along.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ int reflect1(int i, int n) {  // nn.ReflectionPad2d index map
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

struct ThinArgs {
  const float* x;    // [N,H,W,C] pixel pitch xcs
  const float* w;    // wf[9][C][K]
  const float* bias;
  const float* res;  // y layout
  float* y;          // [N,H,W,K] pixel pitch ycs
  int N, H, W, C, K, xcs, ycs, pad_mode, act;
  int flip = 0;  // weights indexed with tap 8 - t: the adjoint of a THIN-INPUT convolution (C_in <= 4) is a thin-output convolution of dy
  // fused input activation: the convolution reads lrelu(x, in_slope) (the Output block: LeakyReLU -> ReflectionPad2d -> conv -> tanh,
  // base_function.py:367-398): applied while the window is staged (forward / weight gradient, C = 32 LDS kernels only); the adjoint
  // multiplies its result by lrelu'(xin) -- one pass over the 1 GB activation each instead of two
  float in_slope = 1.f;
  const float* xin = nullptr;
};
__device__ __forceinline__ float4 lrelu4(float4 v, float s) {
  v.x = v.x > 0.f ? v.x : v.x * s, v.y = v.y > 0.f ? v.y : v.y * s, v.z = v.z > 0.f ? v.z : v.z * s, v.w = v.w > 0.f ? v.w : v.w * s;
  return v;
}
__device__ __forceinline__ float4 lrelu_mask4(float4 g, float4 x, float s) {
  g.x *= x.x > 0.f ? 1.f : s, g.y *= x.y > 0.f ? 1.f : s, g.z *= x.z > 0.f ? 1.f : s, g.w *= x.w > 0.f ? 1.f : s;
  return g;
}

template <int K>
__device__ __forceinline__ void load_w(float (&wr)[9][4][K], const float* __restrict__ w, int C, int c0, bool is_wt = false) {
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < K; ++k)
        wr[t][e][k] = is_wt ? w[((int64_t)t * K + k) * C + c0 + e]   // wt[tap][K][C]
                            : w[((int64_t)t * C + c0 + e) * K + k];  // wf[tap][C][K]
}

// dx[q][c0..c0+3] of one pixel; dy = a.y.  With reflect padding the fold is included: besides u = q + 1 the padded coordinates
// u = 0 (q == 1) and u = H + 1 (q == H - 2) map onto q.
template <int K>
__device__ __forceinline__ float4 thin_dgrad_pixel(const ThinArgs& a, const float (&wr)[9][4][K], int n, int qy, int qx) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int uy[3], ux[3], nuy = 0, nux = 0;
  uy[nuy++] = qy + 1;
  ux[nux++] = qx + 1;
  if (a.pad_mode) {
    if (qy == 1) uy[nuy++] = 0;
    if (qy == a.H - 2) uy[nuy++] = a.H + 1;
    if (qx == 1) ux[nux++] = 0;
    if (qx == a.W - 2) ux[nux++] = a.W + 1;
  }
  for (int iy = 0; iy < nuy; ++iy) {
    for (int ix = 0; ix < nux; ++ix) {
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int py = uy[iy] - ty, px = ux[ix] - tx;  // output pixel whose tap (ty, tx) reads padded position u
          if ((unsigned)py >= (unsigned)a.H || (unsigned)px >= (unsigned)a.W) continue;
          const float* g = a.y + ((int64_t)(n * a.H + py) * a.W + px) * a.ycs;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float gv = g[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(gv, wr[ty * 3 + tx][e][k], acc[e]);
          }
        }
      }
    }
  }
  return make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// the pixels whose gradient contains folded (reflected) contributions: rows 1 and H-2, columns 1 and W-2 of every image
template <int K>
__global__ void __launch_bounds__(256) thin_dgrad_border_kernel(ThinArgs a, float* __restrict__ dx, int64_t total) {
  const int CG = a.C >> 2, per_img = 2 * a.W + 2 * a.H;
  float wr[9][4][K];
  int cur_c0 = -1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % CG);
    int64_t r = i / CG;
    const int b = (int)(r % per_img), n = (int)(r / per_img);
    if (n >= a.N) continue;
    int qy, qx;
    if (b < 2 * a.W) {
      qy = b < a.W ? 1 : a.H - 2;
      qx = b < a.W ? b : b - a.W;
    } else {
      const int bb = b - 2 * a.W;
      qx = bb < a.H ? 1 : a.W - 2;
      qy = bb < a.H ? bb : bb - a.H;
    }
    const int c0 = 4 * j;
    if (c0 != cur_c0) {
      load_w<K>(wr, a.w, a.C, c0, true);
      cur_c0 = c0;
    }
    const int64_t o = ((int64_t)(n * a.H + qy) * a.W + qx) * a.xcs + c0;
    float4 v = thin_dgrad_pixel<K>(a, wr, n, qy, qx);
    if (a.xin) v = lrelu_mask4(v, *reinterpret_cast<const float4*>(a.xin + o), a.in_slope);
    *reinterpret_cast<float4*>(dx + o) = v;
  }
}

template <int K>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(ThinArgs a, float* __restrict__ dwf, float* __restrict__ dbias, int G,
                                                         int64_t total) {
  __shared__ float red[4][9 * 4 * K + K][16];  // [wave][value][channel chunk j]  (G <= 16)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & (G - 1), grp = lane / G, ppw = 64 / G;
  const int c0 = 4 * j;
  float acc[9][4][K], bacc[K];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[t][e][k] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) bacc[k] = 0.f;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t base = wave * ppw; base < total; base += nwaves * ppw) {
    const int64_t p = base + grp;
    if (p >= total) continue;
    const int HW = a.H * a.W;
    const int n = (int)p / HW, rem = (int)p - n * HW;
    const int oy = rem / a.W, ox = rem - oy * a.W;
    float gv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      gv[k] = a.y[p * a.ycs + k];
      bacc[k] += gv[k];
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        int iy = oy + ty - 1, ix = ox + tx - 1;
        if (a.pad_mode) {
          iy = reflect1(iy, a.H);
          ix = reflect1(ix, a.W);
        } else if ((unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) {
          continue;
        }
        const float4 v = *reinterpret_cast<const float4*>(a.x + ((int64_t)(n * a.H + iy) * a.W + ix) * a.xcs + c0);
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < K; ++k) acc[ty * 3 + tx][e][k] = fmaf(xv[e], gv[k], acc[ty * 3 + tx][e][k]);
      }
    }
  }
  // over the wave's pixel groups (lanes that share j), then over the workgroup's waves, then over workgroups
  for (int m = G; m < 64; m <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[t][e][k] += __shfl_xor(acc[t][e][k], m, 64);
#pragma unroll
    for (int k = 0; k < K; ++k) bacc[k] += __shfl_xor(bacc[k], m, 64);
  }
  if (grp == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) red[wid][(t * 4 + e) * K + k][j] = acc[t][e][k];
#pragma unroll
    for (int k = 0; k < K; ++k) red[wid][36 * K + k][j] = bacc[k];
  }
  __syncthreads();
  const int nval = 36 * K;
  for (int i = threadIdx.x; i < nval * G; i += 256) {
    const int v = i / G, jj = i - v * G;
    const float s = red[0][v][jj] + red[1][v][jj] + red[2][v][jj] + red[3][v][jj];
    const int t = v / (4 * K), e = (v / K) & 3, k = v % K;
    atomicAdd(dwf + ((int64_t)t * a.C + 4 * jj + e) * K + k, s);
  }
  if (dbias && threadIdx.x < K) {  // every chunk lane j saw the same pixels: take j = 0
    const int k = threadIdx.x;
    atomicAdd(dbias + k, red[0][36 * K + k][0] + red[1][36 * K + k][0] + red[2][36 * K + k][0] + red[3][36 * K + k][0]);
  }
}


// =====================================================================================
// Matrix-core forms: v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 outer products per instruction; lane l = 4*blk + i
// supplies A_blk[i] and B_blk[i], and register r of lane 4*blk + i accumulates A_blk[r] * B_blk[i] (layout probed on gfx950:
// tools/bench_tools/probe_mfma4.hip) -- so the per-lane quantity (a pixel, a channel) goes on B and the 4-wide one (output channels) on A.  A block is one (pixel group, 4-channel chunk) pair: with C = 32 the 16 blocks
// are 8 chunks x 2 pixel groups, lane = ((g * CQ + q) * 4 + i).  One instruction replaces 4 x K scalar FMAs per lane, the
// operand loads stay the coalesced 16-byte (forward) / 4-byte (gradients) gathers of the scalar kernels above.
// =====================================================================================
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct PixDec {
  int n, y, x;
};
__device__ __forceinline__ PixDec pix_decode(int p, int HW, int W) {
  PixDec d;
  d.n = p / HW;
  const int rem = p - d.n * HW;
  d.y = rem / W;
  d.x = rem - d.y * W;
  return d;
}

// Work decomposition of the three kernels: a wave owns a SEGMENT of one image row (SEG pixels of row (n, y)); the three input rows
// of the 3x3 window are wave-uniform pointers, so a pixel costs no division and no 64-bit address arithmetic.
constexpr int SEG = 128;

struct RowTask {
  int n, y, x0;
};
__device__ __forceinline__ RowTask row_task(int task, int H, int segs) {  // task = ((n * H) + y) * segs + segment
  RowTask r;
  const int row = task / segs;
  r.x0 = (task - row * segs) * SEG;
  r.n = row / H;
  r.y = row - r.n * H;
  return r;
}

// forward: the per-lane quantity is the pixel (B operand), the 4-wide one the output channel (A operand = weights);
// partial sums per channel chunk q are reduced over q by shuffles.
template <int K>
__global__ void __launch_bounds__(256) thin_fwd_mfma_kernel(ThinArgs a, int qbits, int ntasks, int segs) {
  const int lane = threadIdx.x & 63;
  const int CQ = 1 << qbits, i = lane & 3, q = (lane >> 2) & (CQ - 1), g = lane >> (2 + qbits);
  const int ppi = 4 * (16 >> qbits);  // pixels per wave and iteration
  float wr[9][4];                     // A operand: w[t][4q+e][k = i]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[t][e] = i < K ? a.w[((int64_t)(a.flip ? 8 - t : t) * a.C + 4 * q + e) * K + i] : 0.f;
  const int wave = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)(((int64_t)gridDim.x * 256) >> 6);
  for (int task = wave; task < ntasks; task += nwaves) {
    const RowTask rt = row_task(__builtin_amdgcn_readfirstlane(task), a.H, segs);
    const float* rp[3];
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int iy = rt.y + r - 1;
      rok[r] = true;
      if (a.pad_mode) iy = reflect1(iy, a.H);
      else if ((unsigned)iy >= (unsigned)a.H) rok[r] = false, iy = 0;
      rp[r] = a.x + (int64_t)(rt.n * a.H + iy) * a.W * a.xcs + 4 * q;
    }
    const int xend = min(rt.x0 + SEG, a.W);
    float* yrow = a.y + (int64_t)(rt.n * a.H + rt.y) * a.W * a.ycs;
    const float* rrow = a.res ? a.res + (int64_t)(rt.n * a.H + rt.y) * a.W * a.ycs : nullptr;
    for (int xb = rt.x0; xb < xend; xb += ppi) {
      const int x = xb + g * 4 + i;
      const bool live = x < xend;
      int off[3];
      bool cok[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        int ix = (live ? x : xend - 1) + c - 1;
        cok[c] = true;
        if (a.pad_mode) ix = reflect1(ix, a.W);
        else if ((unsigned)ix >= (unsigned)a.W) cok[c] = false, ix = 0;
        off[c] = ix * a.xcs;
      }
      float4 v[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        v[t] = *reinterpret_cast<const float4*>(rp[t / 3] + off[t % 3]);
        if (!(rok[t / 3] && cok[t % 3])) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][0], v[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][1], v[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][2], v[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][3], v[t].w, acc, 0, 0, 0);
      }
      for (int m = 4; m < 4 * CQ; m <<= 1)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], m, 64);
      if (live && q == 0) {
        const int o = x * a.ycs;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float r = acc[k];
          if (a.bias) r += a.bias[k];
          if (rrow) r += rrow[o + k];
          if (a.act == 1) r = tanhf(r);
          else if (a.act == 2) r = fmaxf(r, 0.f);
          yrow[o + k] = r;
        }
      }
    }
  }
}


// Forward with an LDS-staged halo tile (C = 32): a workgroup owns 8 x 32 output pixels, reads the 10 x 34 input window ONCE with
// fully coalesced 16-byte loads (8 consecutive lanes = one pixel's 128 bytes; the texture path is paced per wave instruction and
// the gather of the row-segment kernel above -- 4 pixels x 16 B per MFMA block, 128 B apart -- costs it ~4x as many cycles per byte),
// then every tap comes out of LDS as ds_read_b128 in the MFMA block layout.  Input bytes cross HBM/L2 1.33x instead of 9x.
constexpr int LT_H = 8, LT_W = 32, LT_PITCH = 36;  // pixel pitch in floats (32 channels + 4 pad: spreads the b128 reads over the banks)
template <int K>
__global__ void __launch_bounds__(256) thin_fwd_lds_kernel(ThinArgs a, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float sx[(LT_H + 2) * (LT_W + 2) * LT_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 3, q = (lane >> 2) & 7, g = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % tiles_x;
  b /= tiles_x;
  const int ty = b % tiles_y, n = b / tiles_y;
  const int y0 = ty * LT_H, x0 = tx * LT_W;
  float wr[9][4];  // A operand: w[t][4q+e][k = i]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[t][e] = i < K ? a.w[((int64_t)(a.flip ? 8 - t : t) * a.C + 4 * q + e) * K + i] : 0.f;
  // stage the window: chunk c8 = tid & 7 of window pixel (tid >> 3) + 32 j
  const float* img = a.x + (int64_t)n * a.H * a.W * a.xcs;
  constexpr int NPIX = (LT_H + 2) * (LT_W + 2), NLD = (NPIX + 31) / 32;
  float4 stage[NLD];  // all loads in flight before the first LDS store (a load-store-load chain would pay the latency NLD times)
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int p = (tid >> 3) + 32 * j;
    const int wy = p / (LT_W + 2), wx = p - wy * (LT_W + 2);
    int iy = y0 + wy - 1, ix = x0 + wx - 1;
    if (a.pad_mode) {
      iy = reflect1(iy, a.H);
      ix = reflect1(ix, a.W);
    }
    const bool ok = p < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) stage[j] = *reinterpret_cast<const float4*>(img + ((int64_t)iy * a.W + ix) * a.xcs + 4 * (tid & 7));
  }
  if (a.in_slope != 1.f) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) stage[j] = lrelu4(stage[j], a.in_slope);
  }
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int p = (tid >> 3) + 32 * j;
    if (p < NPIX) *reinterpret_cast<float4*>(sx + p * LT_PITCH + 4 * (tid & 7)) = stage[j];
  }
  __syncthreads();
  // wave `wid` owns tile rows 2*wid, 2*wid+1; an iteration = 8 consecutive pixels of a row (2 groups x 4)
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const int ly = 2 * wid + (it >> 2), lx = (it & 3) * 8 + g * 4 + i;
    const float* base = sx + (ly * (LT_W + 2) + lx) * LT_PITCH + 4 * q;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(base + ((t / 3) * (LT_W + 2) + (t % 3)) * LT_PITCH);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][0], v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][1], v.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][2], v.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][3], v.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int m = 4; m < 32; m <<= 1)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], m, 64);
    const int oy = y0 + ly, ox = x0 + lx;
    if (q < K && oy < a.H && ox < a.W) {  // after the reduction every chunk lane holds the totals: lane q finishes channel q
      const int64_t o = ((int64_t)(n * a.H + oy) * a.W + ox) * a.ycs + q;
      float r = acc[0];
#pragma unroll
      for (int k = 1; k < K; ++k) r = q == k ? acc[k] : r;
      if (a.bias) r += a.bias[q];
      if (a.res) r += a.res[o];
      if (a.act == 1) r = tanhf(r);
      else if (a.act == 2) r = fmaxf(r, 0.f);
      a.y[o] = r;
    }
  }
}


// Weight gradient with the same LDS halo tile: persistent workgroups walk the 8 x 32 tiles, x comes out of LDS (ds_read_b128, lane =
// (pixel group, 4-channel chunk) as in thin_wgrad_kernel), dy as one K-dword load per pixel; the [9][4][K] accumulators live in
// registers across all tiles of the workgroup and are reduced once at the end.
template <int K>
__global__ void __launch_bounds__(256) thin_wgrad_lds_kernel(ThinArgs a, float* __restrict__ dwf, float* __restrict__ dbias, int tiles_x,
                                                             int tiles_y, int ntiles) {
  __shared__ __attribute__((aligned(16))) float sx[(LT_H + 2) * (LT_W + 2) * LT_PITCH];
  float(*red)[8] = reinterpret_cast<float(*)[8]>(sx);  // reused after the last tile: [4 waves * (36 K + K)][8 chunks]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 7, grp = lane >> 3;  // chunk, pixel of the iteration
  float acc[9][4][K], bacc[K];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[t][e][k] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) bacc[k] = 0.f;
  constexpr int NPIX = (LT_H + 2) * (LT_W + 2), NLD = (NPIX + 31) / 32;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int y0 = ty * LT_H, x0 = tx * LT_W;
    const float* img = a.x + (int64_t)n * a.H * a.W * a.xcs;
    float4 stage[NLD];
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int p = (tid >> 3) + 32 * l;
      const int wy = p / (LT_W + 2), wx = p - wy * (LT_W + 2);
      int iy = y0 + wy - 1, ix = x0 + wx - 1;
      if (a.pad_mode) {
        iy = reflect1(iy, a.H);
        ix = reflect1(ix, a.W);
      }
      const bool ok = p < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      stage[l] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) stage[l] = *reinterpret_cast<const float4*>(img + ((int64_t)iy * a.W + ix) * a.xcs + 4 * (tid & 7));
    }
    if (a.in_slope != 1.f) {
#pragma unroll
      for (int l = 0; l < NLD; ++l) stage[l] = lrelu4(stage[l], a.in_slope);
    }
    __syncthreads();  // the previous tile's reads of sx are done
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int p = (tid >> 3) + 32 * l;
      if (p < NPIX) *reinterpret_cast<float4*>(sx + p * LT_PITCH + 4 * (tid & 7)) = stage[l];
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
      const int ly = 2 * wid + (it >> 2), lx = (it & 3) * 8 + grp;
      const int oy = y0 + ly, ox = x0 + lx;
      const bool live = oy < a.H && ox < a.W;
      struct __attribute__((packed, aligned(4))) GK {
        float e[K];
      };
      GK gk;
#pragma unroll
      for (int k = 0; k < K; ++k) gk.e[k] = 0.f;
      if (live) gk = *reinterpret_cast<const GK*>(a.y + ((int64_t)(n * a.H + oy) * a.W + ox) * a.ycs);
#pragma unroll
      for (int k = 0; k < K; ++k) bacc[k] += gk.e[k];
      const float* base = sx + (ly * (LT_W + 2) + lx) * LT_PITCH + 4 * j;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(base + ((t / 3) * (LT_W + 2) + (t % 3)) * LT_PITCH);
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < K; ++k) acc[t][e][k] = fmaf(xv[e], gk.e[k], acc[t][e][k]);
      }
    }
  }
  // over the wave's 8 pixel lanes (same chunk j), then the workgroup's waves through LDS, then workgroups by atomics
  for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[t][e][k] += __shfl_xor(acc[t][e][k], m, 64);
#pragma unroll
    for (int k = 0; k < K; ++k) bacc[k] += __shfl_xor(bacc[k], m, 64);
  }
  __syncthreads();  // sx is free: reuse it for the cross-wave reduction
  constexpr int NV = 36 * K + K;
  if (grp == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) red[wid * NV + (t * 4 + e) * K + k][j] = acc[t][e][k];
#pragma unroll
    for (int k = 0; k < K; ++k) red[wid * NV + 36 * K + k][j] = bacc[k];
  }
  __syncthreads();
  for (int i = tid; i < 36 * K * 8; i += 256) {
    const int v = i >> 3, jj = i & 7;
    const float s_ = red[v][jj] + red[NV + v][jj] + red[2 * NV + v][jj] + red[3 * NV + v][jj];
    const int t = v / (4 * K), e = (v / K) & 3, k = v % K;
    atomicAdd(dwf + ((int64_t)t * a.C + 4 * jj + e) * K + k, s_);
  }
  if (dbias && tid < K) {
    const int v = 36 * K + tid;
    atomicAdd(dbias + tid, red[v][0] + red[NV + v][0] + red[2 * NV + v][0] + red[3 * NV + v][0]);
  }
}

// adjoint with ZERO padding semantics: per-lane quantity = the pixel (B operand = dy of the pixels whose windows cover it), A operand =
// w[t][4q + r][k]; lane (g, q, i) ends with dx[pixel][4q .. 4q+3] in its accumulator: one 16-byte store.  (Reflect padding: the
// caller re-computes the two border rows / columns with thin_dgrad_border_kernel, which knows the fold.)
template <int K>
__global__ void __launch_bounds__(256) thin_dgrad_mfma_kernel(ThinArgs a, float* __restrict__ dx, int qbits, int ntasks, int segs) {
  const int lane = threadIdx.x & 63;
  const int CQ = 1 << qbits, i = lane & 3, q = (lane >> 2) & (CQ - 1), g = lane >> (2 + qbits);
  const int ppi = 4 * (16 >> qbits);
  float wr[9][K];  // A operand: w[t][c = 4q + i][k]   (a.w = wt[tap][K][C])
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < K; ++k) wr[t][k] = a.w[((int64_t)t * K + k) * a.C + 4 * q + i];
  const int wave = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)(((int64_t)gridDim.x * 256) >> 6);
  for (int task = wave; task < ntasks; task += nwaves) {
    const RowTask rt = row_task(__builtin_amdgcn_readfirstlane(task), a.H, segs);
    const float* gp[3];  // dy rows y+1, y, y-1 (tap row ty reads output row y + 1 - ty)
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int py = rt.y + 1 - r;
      rok[r] = (unsigned)py < (unsigned)a.H;
      if (!rok[r]) py = 0;
      gp[r] = a.y + (int64_t)(rt.n * a.H + py) * a.W * a.ycs;
    }
    const int xend = min(rt.x0 + SEG, a.W);
    float* xrow = dx + (int64_t)(rt.n * a.H + rt.y) * a.W * a.xcs + 4 * q;
    for (int xb = rt.x0; xb < xend; xb += ppi) {
      const int x = xb + g * 4 + i;
      const bool live = x < xend;
      float gv[9][K];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        int px = (live ? x : xend - 1) + 1 - c;
        const bool cok = (unsigned)px < (unsigned)a.W;
        if (!cok) px = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          // ONE K-dword load per tap (global_load_dwordx3 for K = 3): the texture path is paced per wave instruction, and the 8
          // lanes of a pixel group all read the same address -- 27 single-dword loads per step were the kernel's bottleneck
          struct __attribute__((packed, aligned(4))) GK {
            float e[K];
          };
          const GK g = *reinterpret_cast<const GK*>(gp[r] + px * a.ycs);
#pragma unroll
          for (int k = 0; k < K; ++k) gv[r * 3 + c][k] = (cok && rok[r]) ? g.e[k] : 0.f;
        }
      }
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[t][k], gv[t][k], acc, 0, 0, 0);
      if (live) {
        float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (a.xin) o = lrelu_mask4(o, *reinterpret_cast<const float4*>(a.xin + (xrow - dx) + (int64_t)x * a.xcs), a.in_slope);
        *reinterpret_cast<float4*>(xrow + (int64_t)x * a.xcs) = o;
      }
    }
  }
}

bool thin_shape_ok(const fmi_conv_desc* d) {
  const int cg = d->C / 4;
  return d->dil <= 1 && (int64_t)d->N * d->H * d->W < (1ll << 31) && d->K >= 1 && d->K <= 4 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->C % 4 == 0 && cg >= 1 &&
         cg <= 16 && (cg & (cg - 1)) == 0 && d->x_cstride % 4 == 0 && d->H >= 3 && d->W >= 3 && d->OH == d->H && d->OW == d->W;
}

}  // namespace

#define THIN_DISPATCH(KERNEL, ...)                                                              \
  switch (d->K) {                                                                               \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;  \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;  \
    case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;  \
    default: hipLaunchKernelGGL((KERNEL<4>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break; \
  }

// 1 if the thin-output kernels take this geometry (same-size 3x3 stride-1 pad-1 convolution, K <= 4, C = 4..64 power of two)
extern "C" int fmi_conv2d_thin_supported(const fmi_conv_desc* d) { return d && thin_shape_ok(d) ? 1 : 0; }

static bool thin_no_lds() {  // debug: FMI_THIN_NO_LDS = never use the LDS halo kernels (read once)
  static const bool v = getenv("FMI_THIN_NO_LDS") != nullptr;
  return v;
}
static bool thin_lrelu_ok(const fmi_conv_desc* d) { return thin_shape_ok(d) && d->C == 32 && !thin_no_lds(); }
extern "C" int fmi_conv2d_thin_lrelu_supported(const fmi_conv_desc* d) { return d && thin_lrelu_ok(d) ? 1 : 0; }

static int thin_fwd_impl(const fmi_conv_desc* d, const float* x, float in_slope, const float* wf, const float* bias, const float* residual,
                         float* y, int act, void* stream) {
  if (!d || !x || !wf || !y || act < 0 || act > 2) return FMI_ERR_BAD_ARG;
  if (!thin_shape_ok(d) || ((uintptr_t)x & 15)) return FMI_ERR_UNSUPPORTED;
  if (in_slope != 1.f && !thin_lrelu_ok(d)) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  ThinArgs a{x, wf, bias, residual, y, d->N, d->H, d->W, d->C, d->K, d->x_cstride, d->y_cstride, d->pad_mode, act};
  a.in_slope = in_slope;
  const int G = d->C / 4;
  const int64_t total = (int64_t)d->N * d->H * d->W;
  if (total >= (1ll << 31)) return FMI_ERR_UNSUPPORTED;
  int qbits = 0;
  while ((1 << qbits) < G) ++qbits;
  if (d->C == 32 && !thin_no_lds()) {  // halo tile through LDS
    const int tiles_x = (d->W + LT_W - 1) / LT_W, tiles_y = (d->H + LT_H - 1) / LT_H;
    const int64_t nb = (int64_t)d->N * tiles_x * tiles_y;
    if (nb < (1ll << 31)) {
      const int grid = (int)nb;
      THIN_DISPATCH(thin_fwd_lds_kernel, a, tiles_x, tiles_y);
      return fmi_launch_status();
    }
  }
  const int segs = (d->W + SEG - 1) / SEG, ntasks = d->N * d->H * segs;
  const int grid = ntasks / 4 > 8192 ? 8192 : (ntasks + 3) / 4;
  THIN_DISPATCH(thin_fwd_mfma_kernel, a, qbits, ntasks, segs);
  return fmi_launch_status();
}

extern "C" int fmi_conv2d_thin_fwd_f32(const fmi_conv_desc* d, const float* x, const float* wf, const float* bias,
                                       const float* residual, float* y, int act, void* stream) {
  return thin_fwd_impl(d, x, 1.f, wf, bias, residual, y, act, stream);
}
extern "C" int fmi_conv2d_thin_lrelu_fwd_f32(const fmi_conv_desc* d, const float* x, float in_slope, const float* wf, const float* bias,
                                             float* y, int act, void* stream) {
  return thin_fwd_impl(d, x, in_slope, wf, bias, nullptr, y, act, stream);
}

/* dx (layout of x) = adjoint of the thin convolution applied to dy; pad_mode = reflect includes the fold of the padded
 * gradient (what fmi_conv2d_dgrad_f32 on the padded extent + fmi_reflect_pad_fold_f32 compute in two passes). */
static int thin_dgrad_impl(const fmi_conv_desc* d, const float* dy, const float* wt, const float* xin, float in_slope, float* dx,
                           void* stream) {
  if (!d || !dy || !wt || !dx) return FMI_ERR_BAD_ARG;
  if (!thin_shape_ok(d) || ((uintptr_t)dx & 15) || ((uintptr_t)xin & 15)) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  ThinArgs a{nullptr, wt, nullptr, nullptr, const_cast<float*>(dy), d->N, d->H, d->W, d->C, d->K, d->x_cstride, d->y_cstride, d->pad_mode, 0};
  a.xin = xin;
  a.in_slope = in_slope;
  int qbits = 0;
  while ((1 << qbits) < d->C / 4) ++qbits;
  {
    const int segs = (d->W + SEG - 1) / SEG, ntasks = d->N * d->H * segs;
    const int grid = ntasks / 4 > 8192 ? 8192 : (ntasks + 3) / 4;
    THIN_DISPATCH(thin_dgrad_mfma_kernel, a, dx, qbits, ntasks, segs);
  }
  if (d->pad_mode) {  // the fold of the reflected border: rows / columns 1 and H-2 / W-2 are re-computed by the kernel that knows it
    const int64_t nb = (int64_t)d->N * (2 * d->W + 2 * d->H) * (d->C / 4);
    const int grid = (int)((nb + 255) / 256);
    THIN_DISPATCH(thin_dgrad_border_kernel, a, dx, nb);
  }
  return fmi_launch_status();
}
extern "C" int fmi_conv2d_thin_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, float* dx, void* stream) {
  return thin_dgrad_impl(d, dy, wt, nullptr, 1.f, dx, stream);
}
extern "C" int fmi_conv2d_thin_lrelu_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* x, float in_slope,
                                               float* dx, void* stream) {
  if (!x) return FMI_ERR_BAD_ARG;
  return thin_dgrad_impl(d, dy, wt, x, in_slope, dx, stream);
}

/* Adjoint of a THIN-INPUT convolution (C_in <= 4, e.g. VGG16's first layer 3 -> 64, whose input gradient d loss / d image the
 * generator needs): dx[N,H,W,C<=4] = thin-output convolution of dy[N,H,W,K] with wt[tap][K][C] and flipped taps (zero padding). */
extern "C" int fmi_conv2d_thin_input_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, float* dx, void* stream) {
  if (!d || !dy || !wt || !dx) return FMI_ERR_BAD_ARG;
  const int cg = d->K / 4;
  if (d->dil > 1 || d->C < 1 || d->C > 4 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 0 || d->K % 4 != 0 || cg < 1 ||
      cg > 16 || (cg & (cg - 1)) != 0 || d->y_cstride % 4 != 0 || d->H < 3 || d->W < 3 || d->OH != d->H || d->OW != d->W ||
      ((uintptr_t)dy & 15) || (int64_t)d->N * d->H * d->W >= (1ll << 31))
    return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  ThinArgs a{dy, wt, nullptr, nullptr, dx, d->N, d->H, d->W, d->K, d->C, d->y_cstride, d->x_cstride, 0, 0};
  a.flip = 1;
  int qbits = 0;
  while ((1 << qbits) < cg) ++qbits;
  const int segs = (d->W + SEG - 1) / SEG, ntasks = d->N * d->H * segs;
  const int grid = ntasks / 4 > 8192 ? 8192 : (ntasks + 3) / 4;
  switch (d->C) {
    case 1: hipLaunchKernelGGL((thin_fwd_mfma_kernel<1>), dim3(grid), dim3(256), 0, st, a, qbits, ntasks, segs); break;
    case 2: hipLaunchKernelGGL((thin_fwd_mfma_kernel<2>), dim3(grid), dim3(256), 0, st, a, qbits, ntasks, segs); break;
    case 3: hipLaunchKernelGGL((thin_fwd_mfma_kernel<3>), dim3(grid), dim3(256), 0, st, a, qbits, ntasks, segs); break;
    default: hipLaunchKernelGGL((thin_fwd_mfma_kernel<4>), dim3(grid), dim3(256), 0, st, a, qbits, ntasks, segs); break;
  }
  return fmi_launch_status();
}

/* dwf[9][C][K] += x^T dy, dbias[k] += sum dy (dbias may be NULL); caller zeroes both. */
static int thin_wgrad_impl(const fmi_conv_desc* d, const float* x, float in_slope, const float* dy, float* dwf, float* dbias, void* stream) {
  if (!d || !x || !dy || !dwf) return FMI_ERR_BAD_ARG;
  if (!thin_shape_ok(d) || ((uintptr_t)x & 15)) return FMI_ERR_UNSUPPORTED;
  if (in_slope != 1.f && !thin_lrelu_ok(d)) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  ThinArgs a{x, nullptr, nullptr, nullptr, const_cast<float*>(dy), d->N, d->H, d->W, d->C, d->K, d->x_cstride, d->y_cstride, d->pad_mode, 0};
  a.in_slope = in_slope;
  const int G = d->C / 4;
  const int64_t total = (int64_t)d->N * d->H * d->W;
  if (d->C == 32 && !thin_no_lds()) {  // halo tile through LDS, persistent workgroups (3 per CU)
    const int tiles_x = (d->W + LT_W - 1) / LT_W, tiles_y = (d->H + LT_H - 1) / LT_H;
    const int64_t nt = (int64_t)d->N * tiles_x * tiles_y;
    if (nt < (1ll << 31)) {
      const int grid = fmi_det() ? 1 : (nt < 768 ? (int)nt : 768);  // reproducible mode: one persistent workgroup walks every tile
      THIN_DISPATCH(thin_wgrad_lds_kernel, a, dwf, dbias, tiles_x, tiles_y, (int)nt);
      return fmi_launch_status();
    }
  }
  const int64_t waves = (total + 64 / G - 1) / (64 / G);
  const int grid = fmi_det() ? 1 : (int)(waves / 4 > 1024 ? 1024 : (waves + 3) / 4);
  THIN_DISPATCH(thin_wgrad_kernel, a, dwf, dbias, G, total);
  return fmi_launch_status();
}
extern "C" int fmi_conv2d_thin_wgrad_f32(const fmi_conv_desc* d, const float* x, const float* dy, float* dwf, float* dbias,
                                         void* stream) {
  return thin_wgrad_impl(d, x, 1.f, dy, dwf, dbias, stream);
}
extern "C" int fmi_conv2d_thin_lrelu_wgrad_f32(const fmi_conv_desc* d, const float* x, float in_slope, const float* dy, float* dwf,
                                               float* dbias, void* stream) {
  return thin_wgrad_impl(d, x, in_slope, dy, dwf, dbias, stream);
}
