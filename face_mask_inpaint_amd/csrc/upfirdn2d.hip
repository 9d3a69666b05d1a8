// upfirdn2d: zero-insert up-sampling, padding / cropping, FIR filtering with the flipped kernel and
// down-sampling of [major][H][W] planes, i.e. the native op of modules/psp/stylegan2/op.
//
// MI355X mapping (HBM-bound: every input and output element should cross HBM once):
//  * one 256-thread workgroup produces a TH x TW = 16 x 64 output tile of one plane; lanes run along x so
//    that global loads and stores are contiguous 256-byte rows per wave;
//  * the input footprint of the tile (with zero fill outside the image) and the flipped FIR taps are staged
//    in LDS once, the up-sampling zeros are never materialised: each output only visits the taps whose
//    up-sampled coordinate lands on a real sample (poly-phase form);
//  * any up / down / kernel size is accepted (the reference's CUDA kernel returns uninitialised memory
//    outside six hard-coded modes, op/upfirdn2d_kernel.cu:172-268).
#include "common.h"
#include <cstdlib>

#define UF_TH 16
#define UF_TW 64

__device__ __forceinline__ int floordiv(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

// element types: fp32, or bf16 stored as uint16_t (arithmetic is fp32 either way, results rounded to nearest even)
__device__ __forceinline__ float uf_ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float uf_ld(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }
__device__ __forceinline__ void uf_st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void uf_st(uint16_t* p, int64_t i, float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;      // quiet NaN keeps its payload bit
  else u += 0x7fffu + ((u >> 16) & 1u);                        // round to nearest even
  p[i] = (uint16_t)(u >> 16);
}

struct UfParams {
  int major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int tile_in_h, tile_in_w, tiles_x, tiles_y;
};

template <class T>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(const T* __restrict__ in, const T* __restrict__ kernel,
                                                        T* __restrict__ out, UfParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sk = smem;                 // [kh][kw] flipped taps
  float* sx = smem + p.kh * p.kw;   // [tile_in_h][tile_in_w]
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x;
  b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int plane = b / p.tiles_y;
  const int oy0 = ty * UF_TH, ox0 = tx * UF_TW;
  // first up-sampled coordinate touched by the tile and the input sample at / before it
  const int iy0 = floordiv(oy0 * p.down_y - p.pad_y0, p.up_y);
  const int ix0 = floordiv(ox0 * p.down_x - p.pad_x0, p.up_x);
  for (int t = tid; t < p.kh * p.kw; t += 256) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
    sk[t] = uf_ld(kernel, (p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx));
  }
  const T* ip = in + (int64_t)plane * p.in_h * p.in_w;
  for (int t = tid; t < p.tile_in_h * p.tile_in_w; t += 256) {
    const int ry = t / p.tile_in_w, rx = t - ry * p.tile_in_w;
    const int iy = iy0 + ry, ix = ix0 + rx;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) v = uf_ld(ip, (int64_t)iy * p.in_w + ix);
    sx[t] = v;
  }
  __syncthreads();
  T* op = out + (int64_t)plane * p.out_h * p.out_w;
  const int lx = tid & (UF_TW - 1);
  const int ox = ox0 + lx;
  if (ox >= p.out_w) return;
  const int ux0 = ox * p.down_x - p.pad_x0;             // up-sampled x of tap kx = 0
  int kx_first = (-ux0) % p.up_x;                       // first kx with (ux0 + kx) % up_x == 0
  if (kx_first < 0) kx_first += p.up_x;
  for (int ly = tid / UF_TW; ly < UF_TH; ly += 256 / UF_TW) {
    const int oy = oy0 + ly;
    if (oy >= p.out_h) break;
    const int uy0 = oy * p.down_y - p.pad_y0;
    int ky_first = (-uy0) % p.up_y;
    if (ky_first < 0) ky_first += p.up_y;
    float acc = 0.f;
    for (int ky = ky_first; ky < p.kh; ky += p.up_y) {
      const int ry = (uy0 + ky) / p.up_y - iy0;  // exact division; may be negative only outside the staged tile -> guarded
      if ((unsigned)ry >= (unsigned)p.tile_in_h) continue;
      for (int kx = kx_first; kx < p.kw; kx += p.up_x) {
        const int rx = (ux0 + kx) / p.up_x - ix0;
        if ((unsigned)rx >= (unsigned)p.tile_in_w) continue;
        acc += sx[ry * p.tile_in_w + rx] * sk[ky * p.kw + kx];
      }
    }
    uf_st(op, (int64_t)oy * p.out_w + ox, acc);
  }
}


// FIR-only fast path (up = down = 1: the Blur of every up-convolution and its gradients -- all of the op's traffic at 1024^2):
// 32 x 64 output tile per workgroup, the (32+KH-1) x (64+KW-1) input window staged once in LDS; a thread owns one column and 8
// rows and slides down it: each staged row is read once (KW values) and feeds the up to KH outputs it touches, taps are
// compile-time.  5.5 LDS reads per output instead of 16 + the modulo arithmetic of the generic poly-phase loop.
#define FIR_TH 32
#define FIR_TW 64
template <class T, int KH, int KW>
__global__ void __launch_bounds__(256) upfirdn2d_fir_kernel(const T* __restrict__ in, const T* __restrict__ kernel, T* __restrict__ out,
                                                            UfParams p) {
  constexpr int IH = FIR_TH + KH - 1, IW = FIR_TW + KW - 1, LDW = IW + 1;
  __shared__ float sx[IH * LDW];
  float kf[KH][KW];  // flipped taps: out[y][x] = sum_{a,b} in[y - pad_y0 + a][x - pad_x0 + b] * kernel[KH-1-a][KW-1-b]
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int b = 0; b < KW; ++b) kf[a][b] = uf_ld(kernel, (KH - 1 - a) * KW + (KW - 1 - b));
  const int tid = threadIdx.x;
  int b_ = blockIdx.x;
  const int tx = b_ % p.tiles_x;
  b_ /= p.tiles_x;
  const int ty = b_ % p.tiles_y;
  const int plane = b_ / p.tiles_y;
  const int oy0 = ty * FIR_TH, ox0 = tx * FIR_TW;
  const int iy0 = oy0 - p.pad_y0, ix0 = ox0 - p.pad_x0;
  const T* ip = in + (int64_t)plane * p.in_h * p.in_w;
  for (int t = tid; t < IH * IW; t += 256) {
    const int ry = t / IW, rx = t - ry * IW;
    const int iy = iy0 + ry, ix = ix0 + rx;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) v = uf_ld(ip, (int64_t)iy * p.in_w + ix);
    sx[ry * LDW + rx] = v;
  }
  __syncthreads();
  const int lx = tid & (FIR_TW - 1), ly0 = (tid >> 6) * 8;
  const int ox = ox0 + lx;
  if (ox >= p.out_w) return;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 8 + KH - 1; ++r) {  // staged row ly0 + r feeds output rows r - a (a = tap row)
    float v[KW];
#pragma unroll
    for (int b = 0; b < KW; ++b) v[b] = sx[(ly0 + r) * LDW + lx + b];
#pragma unroll
    for (int a = 0; a < KH; ++a) {
      const int o = r - a;
      if (o >= 0 && o < 8) {
#pragma unroll
        for (int b = 0; b < KW; ++b) acc[o] = fmaf(v[b], kf[a][b], acc[o]);
      }
    }
  }
  T* op = out + (int64_t)plane * p.out_h * p.out_w;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int oy = oy0 + ly0 + r;
    if (oy < p.out_h) uf_st(op, (int64_t)oy * p.out_w + ox, acc[r]);
  }
}

// Wide FIR path (output rows of a multiple of 4 elements, 16- / 8-byte aligned): 32 x 128 output tile, a thread owns 4 adjacent
// columns x 4 rows.  The scalar form above issues one 4-byte (bf16: 2-byte) load AND store per output element and is bound by
// those instructions -- bf16 took exactly the time of fp32.  Here the window is staged with 16-byte loads (4 fp32 / 8 bf16; the
// addresses are only element- resp. word-aligned: the Blur input is 1025 wide), every staged row is read once for up to KH output
// rows (3 LDS reads per output) and every output row piece leaves as one 16-byte (bf16: 8-byte) store.
#define FW_TH 32
#define FW_TW 128
struct __attribute__((packed, aligned(4))) uf_f4 {
  float v[4];
};
struct __attribute__((packed, aligned(4))) uf_h8 {
  uint32_t w[4];
};
template <class T, int KH, int KW>
__global__ void __launch_bounds__(256) upfirdn2d_firw_kernel(const T* __restrict__ in, const T* __restrict__ kernel, T* __restrict__ out,
                                                             UfParams p) {
  constexpr int IH = FW_TH + KH - 1, IW = FW_TW + KW - 1;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EV = BF ? 8 : 4;  // elements per 16-byte load
  constexpr int VPR = (IW + (BF ? 1 : 0) + EV - 1) / EV;  // 16-byte vectors per window row
  constexpr int LDW = VPR * EV;  // LDS row pitch: whole vectors, so every staged vector is stored with 16-byte ds_write_b128
  __shared__ __attribute__((aligned(16))) float sx[IH * LDW];
  float kf[KH][KW];
#pragma unroll
  for (int a = 0; a < KH; ++a)
#pragma unroll
    for (int b = 0; b < KW; ++b) kf[a][b] = uf_ld(kernel, (KH - 1 - a) * KW + (KW - 1 - b));
  const int tid = threadIdx.x;
  int b_ = blockIdx.x;
  const int tx = b_ % p.tiles_x;
  b_ /= p.tiles_x;
  const int ty = b_ % p.tiles_y;
  const int plane = b_ / p.tiles_y;
  const int oy0 = ty * FW_TH, ox0 = tx * FW_TW;
  const int iy0 = oy0 - p.pad_y0, ix0 = ox0 - p.pad_x0;
  const int64_t pbase = (int64_t)plane * p.in_h * p.in_w;  // element index of the plane
  const int64_t pend = (int64_t)p.major * p.in_h * p.in_w; // one past the last element of the tensor
  // staging: row ry of the window = elements [rowE, rowE + IW) of the flat tensor (where inside the image); vector j of that row
  // starts at the EV-aligned-to-word element (rowE & ~(BF ? 1 : 0)) + EV * j
  // bf16: a window row is staged by one 32-lane half of a wave, so that the realigning shuffle below stays inside the row
  constexpr int LPR = BF ? 32 : VPR;
  static_assert(VPR <= LPR, "lanes per staged row");
  for (int t = tid; t < IH * LPR; t += 256) {
    const int ry = t / LPR, j = t - ry * LPR;
    const int iy = iy0 + ry;
    float* dst = sx + ry * LDW;
    const bool rowok = (unsigned)iy < (unsigned)p.in_h && j < VPR;
    const int64_t rowE = pbase + (int64_t)iy * p.in_w + ix0;        // element of window column 0 (may lie outside the row)
    const int sh = BF ? (int)(rowE & 1) : 0;                         // bf16: loads start on a 4-byte word
    const int64_t e0 = rowE - sh + (int64_t)EV * j;                  // first element of this vector
    float f[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) f[e] = 0.f;
    if (rowok && e0 >= 0 && e0 + EV <= pend) {
      if constexpr (BF) {
        const uf_h8 raw = *reinterpret_cast<const uf_h8*>(in + e0);
#pragma unroll
        for (int q = 0; q < 4; ++q) f[2 * q] = __uint_as_float(raw.w[q] << 16), f[2 * q + 1] = __uint_as_float(raw.w[q] & 0xffff0000u);
      } else {
        const uf_f4 raw = *reinterpret_cast<const uf_f4*>(in + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = raw.v[e];
      }
    } else if (rowok) {  // vector sticks out of the tensor: element-wise
#pragma unroll
      for (int e = 0; e < EV; ++e)
        if (e0 + e >= 0 && e0 + e < pend) f[e] = uf_ld(in, e0 + e);
    }
    if constexpr (BF) {
      // rows whose first element is odd were loaded from one element earlier (word alignment): shift them left by one so that LDS
      // column = window column for every row (the vector of the next lane supplies the last element; at a row's last vector that
      // element lies beyond the window)
      const float nxt = __shfl_down(f[0], 1, 64);
      if (sh) {
#pragma unroll
        for (int e = 0; e < EV - 1; ++e) f[e] = f[e + 1];
        f[EV - 1] = nxt;
      }
    }
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const int ix = ix0 + EV * j + e;  // image column of this element
      if (!rowok || (unsigned)ix >= (unsigned)p.in_w) f[e] = 0.f;
    }
    if (j < VPR) {
      float4* d4 = reinterpret_cast<float4*>(dst + EV * j);
      d4[0] = make_float4(f[0], f[1], f[2], f[3]);
      if constexpr (BF) d4[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
  }
  __syncthreads();
  const int lx = (tid & 31) * 4, ly0 = (tid >> 5) * 4;
  const int ox = ox0 + lx;
  if (ox >= p.out_w) return;
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
#pragma unroll
  for (int r = 0; r < 4 + KH - 1; ++r) {
    float v[8];
    {
      const float4 lo = *reinterpret_cast<const float4*>(sx + (ly0 + r) * LDW + lx);
      const float4 hi = *reinterpret_cast<const float4*>(sx + (ly0 + r) * LDW + lx + 4);
      v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
#pragma unroll
    for (int a = 0; a < KH; ++a) {
      const int o = r - a;
      if (o >= 0 && o < 4) {
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[o][c] = fmaf(v[b + c], kf[a][b], acc[o][c]);
      }
    }
  }
  T* op = out + (int64_t)plane * p.out_h * p.out_w;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = oy0 + ly0 + r;
    if (oy >= p.out_h) break;
    const int64_t o = (int64_t)oy * p.out_w + ox;
    if constexpr (BF) {
      uint16_t h[4];
      for (int c = 0; c < 4; ++c) uf_st(h, c, acc[r][c]);
      *reinterpret_cast<uint2*>(op + o) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    } else {
      *reinterpret_cast<float4*>(op + o) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
}

static bool firw_off() {  // debug: FMI_FIRW_OFF = never use the wide FIR path (read once)
  static const bool v = getenv("FMI_FIRW_OFF") != nullptr;
  return v;
}

template <class T>
static int upfirdn2d_launch(const T* in, const T* kernel, T* out, int major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                            int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  if (!in || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0) return FMI_ERR_BAD_ARG;
  if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return FMI_ERR_BAD_ARG;
  UfParams p;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  const int full_h = in_h * up_y + pad_y0 + pad_y1 - kh, full_w = in_w * up_x + pad_x0 + pad_x1 - kw;
  if (full_h < 0 || full_w < 0) return FMI_ERR_BAD_ARG;
  p.out_h = full_h / down_y + 1;
  p.out_w = full_w / down_x + 1;
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == kw && kh >= 2 && kh <= 4 && p.out_w % 4 == 0 && p.out_w >= 128 &&
      ((uintptr_t)out & (sizeof(T) == 2 ? 7 : 15)) == 0 && ((uintptr_t)in & 3) == 0 && !firw_off()) {
    p.tile_in_h = p.tile_in_w = 0;
    p.tiles_x = (p.out_w + FW_TW - 1) / FW_TW;
    p.tiles_y = (p.out_h + FW_TH - 1) / FW_TH;
    const int64_t nb = (int64_t)major * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
    if (kh == 4) hipLaunchKernelGGL((upfirdn2d_firw_kernel<T, 4, 4>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    else if (kh == 3) hipLaunchKernelGGL((upfirdn2d_firw_kernel<T, 3, 3>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    else hipLaunchKernelGGL((upfirdn2d_firw_kernel<T, 2, 2>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    return fmi_launch_status();
  }
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == kw && kh >= 2 && kh <= 4) {
    p.tile_in_h = p.tile_in_w = 0;
    p.tiles_x = (p.out_w + FIR_TW - 1) / FIR_TW;
    p.tiles_y = (p.out_h + FIR_TH - 1) / FIR_TH;
    const int64_t nb = (int64_t)major * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
    if (kh == 4) hipLaunchKernelGGL((upfirdn2d_fir_kernel<T, 4, 4>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    else if (kh == 3) hipLaunchKernelGGL((upfirdn2d_fir_kernel<T, 3, 3>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    else hipLaunchKernelGGL((upfirdn2d_fir_kernel<T, 2, 2>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, kernel, out, p);
    return fmi_launch_status();
  }
  // input rows touched by a tile: up-sampled extent (TH-1)*down + kh, i.e. at most that / up + 2 samples
  p.tile_in_h = ((UF_TH - 1) * down_y + kh - 1) / up_y + 2;
  p.tile_in_w = ((UF_TW - 1) * down_x + kw - 1) / up_x + 2;
  const size_t lds = sizeof(float) * ((size_t)kh * kw + (size_t)p.tile_in_h * p.tile_in_w);
  if (lds > 64 * 1024) return FMI_ERR_UNSUPPORTED;
  p.tiles_x = (p.out_w + UF_TW - 1) / UF_TW;
  p.tiles_y = (p.out_h + UF_TH - 1) / UF_TH;
  const int64_t blocks = (int64_t)major * p.tiles_x * p.tiles_y;
  if (blocks > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((upfirdn2d_kernel<T>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, in, kernel, out, p);
  return fmi_launch_status();
}

extern "C" int fmi_upfirdn2d_f32(const float* in, const float* kernel, float* out, int major, int in_h, int in_w, int kh,
                                 int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                 int pad_y1, void* stream) {
  return upfirdn2d_launch<float>(in, kernel, out, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
extern "C" int fmi_upfirdn2d_bf16(const uint16_t* in, const uint16_t* kernel, uint16_t* out, int major, int in_h, int in_w, int kh,
                                  int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                  int pad_y1, void* stream) {
  return upfirdn2d_launch<uint16_t>(in, kernel, out, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
}
