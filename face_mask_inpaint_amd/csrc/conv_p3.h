// Convolution forward / adjoint with BOTH operands as ready bf16 piece images (x6.h): no split arithmetic in the consuming waves.
//
// Activation piece image ("p3 image") of an NHWC fp32 tensor with C % 16 == 0, written by the PRODUCER of the tensor
// (fmi_split3_f32, the convolution epilogue ConvEp::y3, the norm / activation kernels):
//     x3[pixel][C / 16][piece 0..2][16]   bf16,  x = x0 + x1 + x2 exactly (split3_pair)
// i.e. 96 contiguous bytes per (pixel, 16-channel group) -- exactly what one row of a 16-deep implicit-GEMM tile needs, so the
// LDS-DMA fetches six 16-byte chunks per row from ONE 96-byte run (the fp32 tile was one 64-byte run per row).
// Weights: the piece images fmi_weight_prepare_f32 writes, wf3 / wt3 = [piece][tap][Cred / 8][Nout][8] (ConvWX3 in gemm_core.h).
//
// LDS image of a stage:
//   A: [BM rows][3 pieces][2 channel halves] 16-byte chunks (96 bytes per row); the channel half of chunk position h is h ^ ((row >> 4) & 1)
//      (XOR on the SOURCE address of the copy and on the reader): ds_read_b128 of 32 consecutive rows at one (piece, half) is then
//      conflict-free in the hardware's 16-lane groups (6 * row mod 16 covers the 8 even 16-byte slots per 8 rows, the swizzle moves
//      the second 8 rows of a group onto the odd ones).
//   B: [3 pieces][2 channel halves][BN] chunks, lane-linear in the output column.
// One 16-deep k-step = 6 ds_read_b128 per 32 rows / columns and 6 MFMAs (v_mfma_f32_32x32x16_bf16) per 32 x 32 block.
#pragma once
#include "gemm_core.h"

#ifndef FMI_P3_EXP
#define FMI_P3_EXP 0  // timing experiments (wrong results): 1 A copies read the zero chunk, 2 B copies read the zero chunk, 4 no MFMAs, 8 no LDS reads in the loop, 16 no barrier
#endif
#ifndef FMI_HOST_EMU
struct ConvK3 {  // A operand: gathered pixels x (tap, 16-channel group) from a p3 image
  const uint16_t* p3;
  ConvGeom g;  // cstride is not used: the pixel pitch of a p3 image is 3 * C elements
  FastDiv dT;  // taps of the launch geometry.  The 16-deep k-steps walk (channel group OUTER, tap INNER): consecutive steps re-read
               // the same channel group of neighbouring pixels while it is still in the XCD's L2 (tap-major order re-fetched the
               // activation tile nine times from beyond L2: 3.6 GB per 8 x 128^2 256 -> 256 launch)
  struct DCtx {
    int64_t boff;  // element offset of (anchor pixel, piece, channel half)
    int ry, rx;    // anchor coordinates; rows beyond M get ry far outside the image
  };
  struct Tile {
    int dy, dx;
    int64_t uoff;
  };
  __device__ __forceinline__ DCtx dprep(int x, int piece, int half) const {
    DCtx d{0, -0x20000000, 0};
    if (x >= g.Mdim()) return d;
    const uint32_t n = fdiv((uint32_t)x, g.dG);
    const uint32_t rem = (uint32_t)x - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    d.ry = (int)gy * g.S + g.dy0;
    d.rx = (int)gx * g.S + g.dx0;
    d.boff = ((int64_t)((int)n * g.IH + d.ry) * g.IW + d.rx) * (3 * g.C) + piece * 16 + half * 8;
    return d;
  }
  // k-step kt (= k0 / 16) -> (channel group, tap)
  __device__ __forceinline__ void decode(int k0, int& cg, int& tp) const {
    const int kt = k0 >> 4;
    cg = (int)fdiv((uint32_t)kt, dT);
    tp = kt - cg * g.ntaps();
  }
  __device__ __forceinline__ Tile tile(int k0) const {
    int cg, tp;
    decode(k0, cg, tp);
    const int i = (int)fdiv((uint32_t)tp, g.dntx), j = tp - i * g.ntx;
    Tile t;
    t.dy = (cg << 4) < g.C ? g.ystep * i : -0x20000000;
    t.dx = g.xstep * j;
    t.uoff = ((int64_t)(g.ystep * i) * g.IW + t.dx) * (3 * g.C) + cg * 48;
    return t;
  }
  // the weight tile of the same k-step: chunk row of (tap, channel group) in wf3 / wt3 = [piece][tap][Cred / 8][Nout][8]
  __device__ __forceinline__ ConvWX3::Tile wtile(const ConvWX3& lb, int k0) const {
    int cg, tp;
    decode(k0, cg, tp);
    if ((cg << 4) >= g.C) return ConvWX3::Tile{nullptr};
    const int i = (int)fdiv((uint32_t)tp, g.dntx), j = tp - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    return ConvWX3::Tile{lb.p3 + ((int64_t)wtap * (g.C >> 3) + cg * 2) * lb.Nout * 8};
  }
  __device__ __forceinline__ const uint16_t* chunk(const DCtx& d, const Tile& t) const {
    const bool ok = (unsigned)(d.ry + t.dy) < (unsigned)g.IH && (unsigned)(d.rx + t.dx) < (unsigned)g.IW;
    return ok ? p3 + d.boff + t.uoff : nullptr;
  }
};

__device__ __forceinline__ void glds16_p3(const void* g, uint32_t dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(dst)
               : "memory");
}

template <class T>
__device__ __forceinline__ void p3_acc_clear(f32x16 (&a)[T::TM][T::TN]) {
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[i][j][r] = 0.f;
}
// dst += src; src = 0
template <class T>
__device__ __forceinline__ void p3_acc_flush(f32x16 (&src)[T::TM][T::TN], f32x16 (&dst)[T::TM][T::TN]) {
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dst[i][j][r] += src[i][j][r];
        src[i][j][r] = 0.f;
      }
}

// ---------------------------------------------------------------------------------------------------------------------------
// generic implicit GEMM: rows = anchors (output pixels / adjoint phase pixels), k = (tap, channel), columns = output channels
// ---------------------------------------------------------------------------------------------------------------------------
// FL: blocked accumulation -- every 512 reduction elements the accumulators are added into a second set and cleared, so that a long
// reduction is a sum of short fp32 chains (what a CPU library's blocked kernels do) instead of ONE chain of up to 4 608 products:
// measured on the whole-pSp fixture, the unsplit sequential form is 3 x further from float64 than the reference's own fp32 run.
template <class T, bool FL = false>
__global__ void __launch_bounds__(256) gemm_p3_kernel(ConvK3 la, ConvWX3 lb, ConvEp ep, int M, int N, int K, int tiles_n, int ksplit, int kchunk) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, BK = 16;
  constexpr int NIA = BM * 6 / 64, NIB = BN * 6 / 64;  // wave instructions (1 KiB each) of the A / B image of a stage
  constexpr int NLA = (NIA + 3) / 4, NLB = (NIB + 3) / 4;
  constexpr int ABYTES = BM * 96, STAGE = (BM + BN) * 96;
  static_assert((BM * 6) % 64 == 0 && (BN * 6) % 64 == 0, "whole wave instructions");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int zs = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_begin = zs * kchunk;
  int k_end = k_begin + kchunk;
  if (k_end > K) k_end = K;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;

  ConvK3::DCtx da[NLA];
  ConvWX3::DCtx db[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int p = (j * 4 + wid) * 64 + lane;  // chunk p of [BM][3][2]
    const int row = p / 6, q = p - row * 6;
    da[j] = la.dprep(p < BM * 6 ? m0 + row : 0x7fffffff, q >> 1, (q & 1) ^ ((row >> 4) & 1));
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = (j * 4 + wid) * 64 + lane;  // chunk p of [3][2][BN]
    const int piece = p / (2 * BN), r = p - piece * (2 * BN), kg = r / BN, x = r - kg * BN;
    db[j] = lb.dprep3(piece < 3 ? piece : 0, kg, piece < 3 ? n0 + x : 0x40000000);
  }
  const int na_w = (NIA - wid + 3) / 4, nb_w = (NIB - wid + 3) / 4;  // wave-uniform: instructions this wave issues per stage

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int k0, int st) __attribute__((always_inline)) {
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE + wid * 1024));
    const uint32_t sb = sa + ABYTES;
    const ConvK3::Tile ta = la.tile(k0);
    const ConvWX3::Tile tb = la.wtile(lb, k0);
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (j >= na_w) break;
      const void* g = la.chunk(da[j], ta);
      if (!g || (FMI_P3_EXP & 1)) g = zchunk;
      glds16_p3(g, sa + j * 4096);
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (j >= nb_w) break;
      const void* g = lb.chunk(db[j], tb);
      if (!g || (FMI_P3_EXP & 2)) g = zchunk;
      glds16_p3(g, sb + j * 4096);
    }
  };
  auto compute = [&](int st) __attribute__((always_inline)) {
    const unsigned char* sa = lds + st * STAGE;
    const unsigned char* sb = sa + ABYTES;
    bf16x8_t pa[T::TM][3], pb[T::TN][3];
#if FMI_P3_EXP & 8
    for (int i = 0; i < T::TM; ++i) for (int pc = 0; pc < 3; ++pc) for (int e = 0; e < 8; ++e) pa[i][pc][e] = (__bf16)(float)(st + i + pc);
    for (int j = 0; j < T::TN; ++j) for (int pc = 0; pc < 3; ++pc) for (int e = 0; e < 8; ++e) pb[j][pc][e] = (__bf16)(float)(st + j + pc);
#else
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const int r = wm + i * 32 + l31;
      const unsigned char* q = sa + r * 96 + ((lh ^ ((r >> 4) & 1)) << 4);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) pa[i][pc] = *reinterpret_cast<const bf16x8_t*>(q + pc * 32);
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) pb[j][pc] = *reinterpret_cast<const bf16x8_t*>(sb + pc * (2 * BN * 16) + (lh * BN + wn + j * 32 + l31) * 16);
#endif
#if FMI_P3_EXP & 4
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) acc[i][j][pc] += (float)pa[i][pc][0] * (float)pb[j][pc][0];
#else
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(pa[i], pb[j], acc[i][j]);
#endif
  };

  const int nt = (k_end - k_begin + BK - 1) / BK;
  if (nt > 0) issue(k_begin, 0);
  int st = 0;
  f32x16 acc2[FL ? T::TM : 1][FL ? T::TN : 1];
  if constexpr (FL) p3_acc_clear<T>(acc2);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of tile t have landed
#if !(FMI_P3_EXP & 16)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    if (t + 1 < nt) issue(k_begin + (t + 1) * BK, st ^ 1);
    compute(st);
    st ^= 1;
    if constexpr (FL)
      if ((t & 31) == 31) p3_acc_flush<T>(acc, acc2);
  }
  if constexpr (FL) p3_acc_flush<T>(acc2, acc);  // acc = acc + acc2 (the tail block joins the flushed ones)
  store_tile<ConvEp, T>(ep, acc, M, N, m0 + wm, n0 + wn, lh, l31);
}

static bool p3_generic_ok(const ConvGeom& g, const void* x3, const void* w3, int Nout) {
  return x3 && w3 && (g.C & 15) == 0 && !g.pad_mode && ((uintptr_t)x3 & 15) == 0 && ((uintptr_t)w3 & 15) == 0 && (Nout & 3) == 0 &&
         (int64_t)g.N * g.IH * g.IW * 3 * g.C < (1ll << 40);
}

static int launch_gemm_p3(const ConvK3& la, const ConvWX3& lb, const ConvEp& ep, int M, int N, int K, int ksplit, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || ksplit <= 0) return FMI_ERR_BAD_ARG;
  int kchunk = (int)ceil_div64(ceil_div64(K, ksplit), 16) * 16;
  ksplit = (int)ceil_div64(K, kchunk);
  if (ksplit > 65535) return FMI_ERR_UNSUPPORTED;
#define P3_LAUNCH(TILE)                                                                                                                        \
  do {                                                                                                                                         \
    const int64_t tm = ceil_div64(M, TILE::BM), tn = ceil_div64(N, TILE::BN);                                                                  \
    if (tm * tn > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;                                                                                    \
    if (fl)                                                                                                                                    \
      hipLaunchKernelGGL((gemm_p3_kernel<TILE, true>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(256), 0, st, la, lb, ep, M, N, K, (int)tn, ksplit, \
                         kchunk);                                                                                                              \
    else                                                                                                                                       \
      hipLaunchKernelGGL((gemm_p3_kernel<TILE>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(256), 0, st, la, lb, ep, M, N, K, (int)tn, ksplit, \
                         kchunk);                                                                                                              \
  } while (0)
  auto wgs = [&](int bm, int bn) { return ceil_div64(M, bm) * ceil_div64(N, bn) * ksplit; };
  // blocked accumulation (a second accumulator set: ~64 registers, an occupancy step for the 128 x 128 tile) where one workgroup adds more
  // than ~600 products in a row AND the run asks for it: the reproducible mode, where no split reduction blocks the sum for us
  const bool fl = kchunk > 640 && (fmi_det() || fmi_blocked_acc());
  static const int tile_dbg = getenv("FMI_P3_TILE") ? atoi(getenv("FMI_P3_TILE")) : 0;  // experiment: force a tile
  if (tile_dbg == 1) { P3_LAUNCH(Tile128x128); return fmi_launch_status(); }
  if (tile_dbg == 2) { P3_LAUNCH(Tile64x128); return fmi_launch_status(); }
  if (tile_dbg == 3) { P3_LAUNCH(Tile128x64); return fmi_launch_status(); }
  if (N <= 32) {
    P3_LAUNCH(Tile128x32);
  } else if (N <= 64) {
    if (M > 64 && wgs(128, 64) >= 384) P3_LAUNCH(Tile128x64);
    else P3_LAUNCH(Tile64x64);
  } else {
    if (M > 64 && wgs(128, 128) >= 800) P3_LAUNCH(Tile128x128);
    else if (M > 32 && wgs(64, 128) >= 384) P3_LAUNCH(Tile64x128);
    else if (M > 64 && wgs(32, 128) < 256) P3_LAUNCH(Tile64x128);
    else P3_LAUNCH(Tile32x128);
  }
#undef P3_LAUNCH
  return fmi_launch_status();
}
#endif

#ifndef FMI_HOST_EMU
// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 zero-padded convolution (forward, and the adjoint with flipped taps) with TAP REUSE along x, both operands as pieces.
//
// What bounds the piece-image kernels is the L2 -> LDS path (~24-28 B/clk/CU): a 128 x 128 generic tile needs 24 KB per 768 matrix
// cycles = 32 B/clk/CU at the matrix peak (measured: the copies alone run at the equivalent of 320 TFLOP/s, copies + MFMAs at 180).
// Here a step is one (16-channel group, kernel row): the A image holds BM + 2 consecutive pixels of the flattened [N][H][W] order at
// image row y + ky - 1 and serves the three kx taps from fragment rows shifted by 0 / 1 / 2 (conv3x3.h); with a 256 x 128 tile on
// 8 waves (2 per SIMD, one workgroup per CU) a step moves 60 KB for 4608 matrix cycles per SIMD = 13 B/clk/CU.
//   LDS stage: A [RA rows][3 pieces][2 halves] chunks (half XOR (row >> 4) & 1, as above), B [3 kx][3 pieces][2 halves][BN] chunks.
// ---------------------------------------------------------------------------------------------------------------------------
template <int WM_, int WN_, int TM_, int TN_>
struct P3Tile {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, NW = WM * WN;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
};
struct C3P3Args {
  const uint16_t* x3;  // p3 image of the gathered tensor [N][H][W][C]
  const uint16_t* w3;  // [3][9][C / 8][Nout][8]
  int N, H, W, C, Nout, flip;
  FastDiv dW, dHW;
};

template <class T, bool FL = false>
__global__ void __launch_bounds__(T::NW * 64) conv3x3_p3_kernel(C3P3Args a, ConvEp ep, int M, int tiles_n, int ksplit, int it_chunk) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, NW = T::NW;
  constexpr int NIA = ((BM + 2) * 6 + 63) / 64;  // wave instructions (1 KiB) of the A image
  constexpr int NIB = 18 * BN / 64;              // ... of the three weight tiles
  constexpr int NLA = (NIA + NW - 1) / NW, NLB = (NIB + NW - 1) / NW;
  constexpr int ABYTES = NIA * 1024, STAGE = ABYTES + NIB * 1024;
  static_assert((18 * BN) % 64 == 0, "whole wave instructions");
  static_assert(2 * STAGE <= 160 * 1024, "two stages in LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int zs = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nit_all = 3 * (a.C >> 4);
  const int it_begin = zs * it_chunk;
  int it_end = it_begin + it_chunk;
  if (it_end > nit_all) it_end = nit_all;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;
  const int HW = a.H * a.W;

  // A copy slots: chunk p of [RA][3][2]; row r <-> flattened pixel m0 - 1 + r
  int64_t abase[NLA];
  int ay[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int p = (j * NW + wid) * 64 + lane;
    const int r = p / 6, q = p - r * 6;
    const int an = m0 - 1 + r;
    ay[j] = -0x20000000;
    abase[j] = 0;
    if (an >= 0 && an < M && r < BM + 2) {
      const uint32_t n = fdiv((uint32_t)an, a.dHW);
      const uint32_t rem = (uint32_t)an - n * (uint32_t)HW;
      ay[j] = (int)fdiv(rem, a.dW);
      abase[j] = (int64_t)an * (3 * a.C) + (q >> 1) * 16 + (((q & 1) ^ ((r >> 4) & 1)) << 3);
    }
  }
  // B copy slots: chunk p of [3 kx][3 pieces][2 halves][BN]
  int boff[NLB], bkx[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = (j * NW + wid) * 64 + lane;
    const int per_tap = 6 * BN;
    const int kx = p / per_tap, q = p - kx * per_tap;
    const int piece = q / (2 * BN), r = q - piece * (2 * BN), kg = r / BN, n = r - kg * BN;
    bkx[j] = kx | (piece << 2);
    boff[j] = (n0 + n < a.Nout && kx < 3) ? (kg * a.Nout + n0 + n) * 8 : -1;
  }
  const int na_w = (NIA - wid + NW - 1) / NW, nb_w = (NIB - wid + NW - 1) / NW;
  bool xl[T::TM], xr[T::TM];
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
    const uint32_t an = (uint32_t)(m0 + wm + i * 32 + l31);
    const uint32_t rem = an - fdiv(an, a.dHW) * (uint32_t)HW;
    const uint32_t x = rem - fdiv(rem, a.dW) * (uint32_t)a.W;
    xl[i] = x == 0;
    xr[i] = x == (uint32_t)a.W - 1;
  }

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int64_t wpiece = (int64_t)9 * a.C * a.Nout;
  auto issue = [&](int it, int st) __attribute__((always_inline)) {
    const int cg = it / 3, ky = it - 3 * cg;  // wave-uniform: channel group outer, kernel row inner (L2 reuse of the activation rows)
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE + wid * 1024));
    const uint32_t sb = sa + ABYTES;
    const int64_t aoff = (int64_t)(ky - 1) * a.W * (3 * a.C) + cg * 48;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (j >= na_w) break;
      const bool ok = (unsigned)(ay[j] + ky - 1) < (unsigned)a.H;
      const void* g = ok ? (const void*)(a.x3 + abase[j] + aoff) : (const void*)zchunk;
      glds16_p3(g, sa + (uint32_t)(j * NW * 1024));
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (j >= nb_w) break;
      const int tap = ky * 3 + (bkx[j] & 3);
      const void* g = boff[j] >= 0 ? (const void*)(a.w3 + (int64_t)(bkx[j] >> 2) * wpiece + ((int64_t)(a.flip ? 8 - tap : tap) * (a.C >> 3) + cg * 2) * a.Nout * 8 + boff[j])
                                   : (const void*)zchunk;
      glds16_p3(g, sb + (uint32_t)(j * NW * 1024));
    }
  };
  auto compute = [&](int st) __attribute__((always_inline)) {
    const unsigned char* sa = lds + st * STAGE;
    const unsigned char* sb = sa + ABYTES;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      bf16x8_t pa[T::TM][3], pb[T::TN][3];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const int r = wm + i * 32 + l31 + kx;
        const unsigned char* q = sa + r * 96 + ((lh ^ ((r >> 4) & 1)) << 4);
        const bool zero = (kx == 0 && xl[i]) || (kx == 2 && xr[i]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          u32x4_t v = *reinterpret_cast<const u32x4_t*>(q + pc * 32);
          if (kx != 1) v = zero ? u32x4_t{0u, 0u, 0u, 0u} : v;
          pa[i][pc] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          pb[j][pc] = *reinterpret_cast<const bf16x8_t*>(sb + ((kx * 3 + pc) * 2 * BN + lh * BN + wn + j * 32 + l31) * 16);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(pa[i], pb[j], acc[i][j]);
    }
  };

  if (it_begin < it_end) issue(it_begin, 0);
  int st = 0, since = 0;
  f32x16 acc2[FL ? T::TM : 1][FL ? T::TN : 1];
  if constexpr (FL) p3_acc_clear<T>(acc2);
  for (int it = it_begin; it < it_end; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 1 < it_end) issue(it + 1, st ^ 1);
    compute(st);
    st ^= 1;
    if constexpr (FL) {  // a step adds 48 products per output (three taps x 16 channels): a block of 10 steps = 480
      if (++since == 10) {
        p3_acc_flush<T>(acc, acc2);
        since = 0;
      }
    }
  }
  if constexpr (FL) p3_acc_flush<T>(acc2, acc);
  store_tile<ConvEp, T>(ep, acc, M, a.Nout, m0 + wm, n0 + wn, lh, l31);
}

using C3P3_256x128 = P3Tile<4, 2, 2, 2>;  // 8 waves, 64 x 64 per wave
using C3P3_256x64 = P3Tile<4, 2, 2, 1>;   // 8 waves, 64 x 32 per wave
using C3P3_128x128 = P3Tile<2, 2, 2, 2>;  // 4 waves
using C3P3_128x64 = P3Tile<4, 1, 1, 2>;   // 4 waves, every wave spans both column tiles

static bool conv3x3_p3_eligible(const void* x3, const void* w3, int C, int Nout, int64_t pixels) {
  return x3 && w3 && (C & 15) == 0 && (Nout & 3) == 0 && ((uintptr_t)x3 & 15) == 0 && ((uintptr_t)w3 & 15) == 0 && pixels < (1ll << 30);
}

static int launch_conv3x3_p3(const C3P3Args& a, const ConvEp& ep, int M, int ksplit, hipStream_t st) {
  const int nit = 3 * (a.C >> 4);
  if (ksplit > nit) ksplit = nit;
  const int it_chunk = (nit + ksplit - 1) / ksplit;
  ksplit = (nit + it_chunk - 1) / it_chunk;
#define C3P3_LAUNCH(TILE)                                                                                                                          \
  do {                                                                                                                                             \
    const int64_t tm = ceil_div64(M, TILE::BM), tn = ceil_div64(a.Nout, TILE::BN);                                                                 \
    if (fl)                                                                                                                                        \
      hipLaunchKernelGGL((conv3x3_p3_kernel<TILE, true>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(TILE::NW * 64), 0, st, a, ep, M, (int)tn, \
                         ksplit, it_chunk);                                                                                                        \
    else                                                                                                                                           \
      hipLaunchKernelGGL((conv3x3_p3_kernel<TILE>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(TILE::NW * 64), 0, st, a, ep, M, (int)tn,    \
                         ksplit, it_chunk);                                                                                                        \
  } while (0)
  auto wgs = [&](int bm, int bn) { return ceil_div64(M, bm) * ceil_div64(a.Nout, bn) * ksplit; };
  static const int tile_dbg = getenv("FMI_C3P3_TILE") ? atoi(getenv("FMI_C3P3_TILE")) : 0;  // experiment: force a tile
  const int N = a.Nout;
  // blocked accumulation: the eight-wave tiles hold one workgroup per CU (LDS) at 150 of 256 registers -- the second accumulator set is
  // free there, so every long reduction takes it; the four-wave tiles only on request (occupancy)
  const bool long_red = it_chunk > 13;
  int pick;
  if (tile_dbg) pick = tile_dbg;
  else if (N > 64) pick = wgs(256, 128) >= 256 ? 1 : 3;
  else pick = wgs(256, 64) >= 256 ? 2 : 4;
  const bool fl = long_red && (pick <= 2 || fmi_det() || fmi_blocked_acc());
  if (pick == 1) C3P3_LAUNCH(C3P3_256x128);
  else if (pick == 2) C3P3_LAUNCH(C3P3_256x64);
  else if (pick == 3) C3P3_LAUNCH(C3P3_128x128);
  else C3P3_LAUNCH(C3P3_128x64);
#undef C3P3_LAUNCH
  return fmi_launch_status();
}
#endif

#ifndef FMI_HOST_EMU
// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient with both operands as pieces: dwf[tap][c][k] += sum over pixels x[pixel + tap][c] * dy[pixel][k].
// The reduction index (pixels) is the SLOW index of both NHWC piece images, so the MFMA operands come from ds_read_b64_tr_b16 (the
// hardware 4 x 16 transpose), as in the bf16 weight gradient (conv_bf16.hip): LDS image of one (operand, piece, 32-row group) =
// [16 pixels][32 rows] bf16 = 1 KiB = ONE wave instruction of the copy (lane = pixel * 4 + 8-channel chunk), so all copies of a thread
// share one pixel decode per stage; a stage is one 16-pixel k-step.  Pixel splits over workgroups meet through fp32 atomics; a split is
// pinned to one XCD (its tiles re-read the same pixels through that XCD's L2).
// ---------------------------------------------------------------------------------------------------------------------------
typedef short p3_s16x4 __attribute__((ext_vector_type(4)));
struct WgP3Args {
  const uint16_t* x3;
  const uint16_t* dy3;
  float* dwf;
  ConvGeom g;  // forward geometry: anchors = output pixels, all kh*kw taps
  int Kout, Mrows, P, kchunk;
  int tiles, ksplit, xcd_splits;
};

template <class T, bool FL = false>
__global__ void __launch_bounds__(T::NW * 64) wgrad_p3_kernel(WgP3Args a, int tiles_n) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, BK = 16, NW = T::NW;
  constexpr int GA = BM / 32, GB = BN / 32;       // 32-row groups per operand tile
  constexpr int NI = 3 * (GA + GB);               // wave instructions per stage
  constexpr int NL = (NI + NW - 1) / NW;
  constexpr int STAGE = NI * 1024;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int lid, split;
  if (a.xcd_splits) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = idx / a.tiles;
    lid = idx - grp * a.tiles;
    split = grp * 8 + xcd;
    if (split >= a.ksplit) return;
  } else {
    lid = xcd_remap(blockIdx.x, gridDim.x);
    split = blockIdx.y;
  }
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;
  const ConvGeom& g = a.g;
  const int k_begin = split * a.kchunk;
  int k_end = k_begin + a.kchunk;
  if (k_end > a.P) k_end = a.P;

  // copy slot j of this wave = instruction j * NW + wid of [A: 3 pieces x GA groups][B: 3 pieces x GB groups]
  const int pl = lane >> 2, cq = lane & 3;  // pixel of the k-step, 8-channel chunk of the 32-row group
  int s_dy[NL], s_dx[NL];                   // A slots: tap displacement (dy far outside = no copy); B slots: dy = 0x40000000 marks them
  int64_t s_off[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int ii = j * NW + wid;
    s_dy[j] = -0x20000000, s_dx[j] = 0, s_off[j] = 0;
    if (ii < 3 * GA) {
      const int piece = ii / GA, gr = ii - piece * GA;
      const int row = m0 + 32 * gr;  // (tap, channel) row: a group of 32 rows lies inside one tap (C % 32 == 0)
      if (row < a.Mrows) {
        const int t = (int)fdiv((uint32_t)row, g.dC);
        const int i = (int)fdiv((uint32_t)t, g.dntx), jx = t - i * g.ntx;
        const int ch = row - t * g.C + 8 * cq;
        s_dy[j] = g.dy0 + g.ystep * i;
        s_dx[j] = g.dx0 + g.xstep * jx;
        s_off[j] = ((int64_t)s_dy[j] * g.IW + s_dx[j]) * (3 * g.C) + (ch >> 4) * 48 + piece * 16 + ((ch >> 3) & 1) * 8;
      }
    } else if (ii < NI) {
      const int q = ii - 3 * GA, piece = q / GB, gr = q - piece * GB;
      const int col = n0 + 32 * gr + 8 * cq;
      s_dy[j] = col < a.Kout ? 0x40000000 : -0x20000000;
      s_off[j] = (col >> 4) * 48 + piece * 16 + ((col >> 3) & 1) * 8;
    }
  }
  const int n_w = (NI - wid + NW - 1) / NW;

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int k0, int st) __attribute__((always_inline)) {
    const uint32_t s0 = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE + wid * 1024));
    const int pix = k0 + pl;
    const bool pv = pix < k_end;
    const uint32_t n = fdiv((uint32_t)pix, g.dG);
    const uint32_t rem = (uint32_t)pix - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    const int iy0 = (int)gy * g.S, ix0 = (int)gx * g.S;
    const int64_t xb = ((int64_t)((int)n * g.IH + iy0) * g.IW + ix0) * (3 * g.C);
    const int64_t yb = (int64_t)pix * (3 * a.Kout);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j >= n_w) break;
      const void* gp;
      if (s_dy[j] == 0x40000000) {
        gp = pv ? (const void*)(a.dy3 + yb + s_off[j]) : (const void*)zchunk;
      } else {
        const bool ok = pv && (unsigned)(iy0 + s_dy[j]) < (unsigned)g.IH && (unsigned)(ix0 + s_dx[j]) < (unsigned)g.IW;
        gp = ok ? (const void*)(a.x3 + xb + s_off[j]) : (const void*)zchunk;
      }
      glds16_p3(gp, s0 + (uint32_t)(j * NW * 1024));
    }
  };
  // transposed fragment read of one [16 pixels][32 rows] image: lane 4q+p of a 16-lane group addresses pixel row q, rows 4p..4p+3 of
  // the group's 16; the second read covers pixels +4 (256 bytes on)
  const int i16 = lane & 15;
  const uint32_t lbase = (uint32_t)((8 * lh + (i16 >> 2)) * 64 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3));
  auto frag = [&](uint32_t img) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) p3_s16x4* lp;
    const uint32_t ad = img + lbase;
    union {
      p3_s16x4 h[2];
      bf16x8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)ad);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ad + 256u));
    return u.v;
  };
  auto compute = [&](int st) __attribute__((always_inline)) {
    const uint32_t sa = lds0 + (uint32_t)(st * STAGE), sb = sa + 3 * GA * 1024;
    bf16x8_t pa[T::TM][3], pb[T::TN][3];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) pa[i][pc] = frag(sa + (uint32_t)((pc * GA + wm / 32 + i) * 1024));
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) pb[j][pc] = frag(sb + (uint32_t)((pc * GB + wn / 32 + j) * 1024));
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(pa[i], pb[j], acc[i][j]);
  };
  const int nt = (k_end - k_begin + BK - 1) / BK;
  if (nt > 0) issue(k_begin, 0);
  int st = 0;
  f32x16 acc2[FL ? T::TM : 1][FL ? T::TN : 1];
  if constexpr (FL) p3_acc_clear<T>(acc2);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < nt) issue(k_begin + (t + 1) * BK, st ^ 1);
    compute(st);
    st ^= 1;
    if constexpr (FL)
      if ((t & 31) == 31) p3_acc_flush<T>(acc, acc2);
  }
  if constexpr (FL) p3_acc_flush<T>(acc2, acc);
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= a.Mrows) continue;
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int col = n0 + wn + j * 32 + l31;
        if (col < a.Kout) atomicAdd(a.dwf + (int64_t)row * a.Kout + col, acc[i][j][r]);
      }
    }
  }
}

using WG3_128x128 = P3Tile<2, 2, 2, 2>;
using WG3_128x64 = P3Tile<2, 2, 2, 1>;
using WG3_128x32 = P3Tile<4, 1, 1, 1>;
using WG3_256x128 = P3Tile<4, 2, 2, 2>;  // 8 waves

// ---------------------------------------------------------------------------------------------------------------------------
// 3 x 3 stride-1 zero-padded weight gradient with TAP REUSE along x.  The kernel above stages x once per tap: a 128 x 128 tile moves
// 24 KB per 16-pixel step for 768 matrix cycles = 32 B/clk/CU, more than the L2 -> LDS path delivers (24 - 28).  Here a tile is one
// kernel ROW: its rows are (kx = 0..2) x (32 GA channels), its A image holds the 16 pixels of the step PLUS ONE on either side, at
// image row y + ky - 1, and the fragments of the three kx taps are read from it at pixel offsets -1 / 0 / +1 (a transposed read
// addresses its four pixel rows per lane, so the two outer pixels may live in a separate "halo" strip that one partial wave
// instruction fills).  GA = 2, 128 columns: 19 KB per step for 1152 matrix cycles per SIMD = 17 B/clk/CU.  Measured against the
// kernel above (one box): 256 -> 128 at 128^2 175 vs 153 TFLOP/s, 128 -> 64 at 256^2 154 vs 125, 128 -> 128 at 64^2 123 vs 98.  Two
// workgroups per CU are essential (launch bounds: 268 -> 256 registers; at one workgroup per CU it LOSES to the kernel above).
//   What the shifted rows must not see is the pixel of the neighbouring image row where x - 1 / x + 1 leaves the image: the pixel is
// the REDUCTION index here (an element of the fragment registers, the same for every lane), so the one pixel of a step with x = 0
// (kx = 0) or x = W - 1 (kx = 2) -- at most one each, W >= 16 -- is cleared in the A fragments by a wave-uniform mask.
//   LDS stage: [3 GA images (piece, 32-channel group)][16 pixels][32 ch] | halo [3 GA][2][32 ch] | B [3 GB images][16 pixels][32 cols].
// ---------------------------------------------------------------------------------------------------------------------------
struct Wg3x3Args {
  const uint16_t* x3;
  const uint16_t* dy3;
  float* dwf;
  int H, W, C, Kout, P, kchunk, ksplit, tiles, xcd_splits;
  FastDiv dW, dHW;
};

template <int GA, int TN, bool FL = false>
__global__ void __launch_bounds__(256, FL ? 1 : 2) wgrad3x3_p3_kernel(Wg3x3Args a, int tiles_n) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int NW = 4, WN = 4 / GA, BN = WN * TN * 32, GB = BN / 32, BK = 16;
  constexpr int NIA = 3 * GA, NIB = 3 * GB, NI = NIA + NIB + 1;  // main A images, B images, one halo instruction
  constexpr int NL = (NI + NW - 1) / NW;
  constexpr int HALO = NIA * 1024, BOFF = HALO + 1024, STAGE = BOFF + NIB * 1024;  // the halo strip takes 3 GA * 128 bytes of its KiB
  static_assert(NIA * 8 <= 64, "one wave instruction fills the halo strip");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int lid, split;
  if (a.xcd_splits) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = idx / a.tiles;
    lid = idx - grp * a.tiles;
    split = grp * 8 + xcd;
    if (split >= a.ksplit) return;
  } else {
    lid = xcd_remap(blockIdx.x, gridDim.x);
    split = blockIdx.y;
  }
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int cgs = a.C / (32 * GA);
  const int ky = tile_m / cgs, cb = (tile_m - ky * cgs) * 32 * GA;  // kernel row, first channel
  const int n0 = tile_n * BN;
  const int wm = wid / WN, wn = (wid % WN) * TN * 32;              // this wave: channel group wm, columns wn ..
  const int k_begin = split * a.kchunk;
  int k_end = k_begin + a.kchunk;
  if (k_end > a.P) k_end = a.P;
  const int HW = a.H * a.W;

  // copy slots: instruction ii = j * NW + wid of [A main: 3 GA][B: 3 GB][halo].  Branch-free per step: a slot is (base, pixel pitch, offset,
  // which of the thread's two tracked pixels, whether the image row must be inside) -- the thread tracks (x, y) of its main pixel
  // k0 + pl and of its halo pixel (k0 - 1 or k0 + 16 by lane) incrementally, 16 pixels a step, W >= 16
  const int pl = lane >> 2, cq = lane & 3;
  const uint16_t* s_base[NL];
  int s_pitch[NL];
  int64_t s_off[NL];
  bool s_on[NL], s_halo[NL], s_row[NL];
  uint32_t s_dst[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int ii = j * NW + wid;
    s_base[j] = a.x3, s_pitch[j] = 3 * a.C, s_off[j] = 0, s_on[j] = false, s_halo[j] = false, s_row[j] = false;
    uint32_t dst = 0;
    if (ii < NIA) {
      const int piece = ii / GA, ch = cb + 32 * (ii - piece * GA) + 8 * cq;
      s_on[j] = true, s_row[j] = true;
      s_off[j] = (int64_t)(ky - 1) * a.W * (3 * a.C) + (ch >> 4) * 48 + piece * 16 + ((ch >> 3) & 1) * 8;
      dst = (uint32_t)(ii * 1024);
    } else if (ii < NIA + NIB) {
      const int q = ii - NIA, piece = q / GB, col = n0 + 32 * (q - piece * GB) + 8 * cq;
      s_base[j] = a.dy3, s_pitch[j] = 3 * a.Kout;
      s_on[j] = col < a.Kout;
      s_off[j] = (col >> 4) * 48 + piece * 16 + ((col >> 3) & 1) * 8;
      dst = (uint32_t)(BOFF + q * 1024);
    } else if (ii == NIA + NIB) {
      const int img = lane >> 3, piece = img / GA, ch = cb + 32 * (img - piece * GA) + 8 * cq;  // lane = img * 8 + side * 4 + cq
      s_on[j] = img < NIA, s_halo[j] = true, s_row[j] = true;
      s_off[j] = (int64_t)(ky - 1) * a.W * (3 * a.C) + (ch >> 4) * 48 + piece * 16 + ((ch >> 3) & 1) * 8;
      dst = (uint32_t)HALO;
    }
    s_dst[j] = __builtin_amdgcn_readfirstlane(dst);
  }
  const int n_w = (NI - wid + NW - 1) / NW;

  f32x16 acc[3][TN];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // tracked pixels: main = k_begin + pl, halo = k_begin - 1 / + 16; (x, y) of each, y = image row of the OUTPUT pixel
  const int hdp = ((lane >> 2) & 1) ? 16 : -1;
  int pm = k_begin + pl, ph = k_begin + hdp;
  int xm, ym, xh, yh;
  {
    const uint32_t n = fdiv((uint32_t)pm, a.dHW), rem = (uint32_t)pm - n * (uint32_t)HW;
    ym = (int)fdiv(rem, a.dW), xm = (int)rem - ym * a.W;
    const int phc = ph < 0 ? 0 : ph;
    const uint32_t n2 = fdiv((uint32_t)phc, a.dHW), rem2 = (uint32_t)phc - n2 * (uint32_t)HW;
    yh = (int)fdiv(rem2, a.dW), xh = (int)rem2 - yh * a.W;
    if (ph < 0) xh = a.W - 1, yh = a.H - 1;  // "pixel -1": one step on it becomes pixel 15 of the first row
  }
  auto advance = [&]() __attribute__((always_inline)) {
    pm += BK, ph += BK;
    xm += BK, xh += BK;
    if (xm >= a.W) { xm -= a.W; ym = ym + 1 == a.H ? 0 : ym + 1; }
    if (xh >= a.W) { xh -= a.W; yh = yh + 1 == a.H ? 0 : yh + 1; }
  };
  auto issue = [&](int st) __attribute__((always_inline)) {  // copies the step the tracked pixels stand on
    const uint32_t s0 = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE));
    const bool vm = pm < k_end, vh = ph >= 0 && ph < a.P;
    const bool rm = (unsigned)(ym + ky - 1) < (unsigned)a.H, rh = (unsigned)(yh + ky - 1) < (unsigned)a.H;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j >= n_w) break;
      const int pix = s_halo[j] ? ph : pm;
      const bool ok = s_on[j] && (s_halo[j] ? vh : vm) && (!s_row[j] || (s_halo[j] ? rh : rm));
      const void* gp = ok ? (const void*)(s_base[j] + (int64_t)pix * s_pitch[j] + s_off[j]) : (const void*)zchunk;
      glds16_p3(gp, s0 + s_dst[j]);
    }
  };
  // transposed fragment reads.  Pixel row p of an A image (p = -1 .. 16): main image [16][64 B], p = -1 / 16 in the halo strip.  All
  // per-lane offsets are loop constants; the stage and the image enter as immediates (the step loop is unrolled over the two stages)
  const int i16 = lane & 15;
  const uint32_t lcol = (uint32_t)(32 * ((lane >> 4) & 1) + 8 * (i16 & 3));
  const int prow = 8 * lh + (i16 >> 2);
  typedef __attribute__((address_space(3))) p3_s16x4* lp;
  uint32_t offA[3][2], offA0[3], offA2[3];  // [kx][read]; the two reads that touch the halo strip: per piece
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < 2; ++r) offA[kx][r] = lds0 + (uint32_t)(wm * 1024 + (prow + 4 * r + kx - 1) * 64) + lcol;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    offA0[pc] = prow - 1 < 0 ? lds0 + (uint32_t)(HALO + (pc * GA + wm) * 128) + lcol : offA[0][0] + (uint32_t)(pc * GA * 1024);
    offA2[pc] = prow + 5 > 15 ? lds0 + (uint32_t)(HALO + (pc * GA + wm) * 128 + 64) + lcol : offA[2][1] + (uint32_t)(pc * GA * 1024);
  }
  const uint32_t offB = lds0 + (uint32_t)(BOFF + (wn / 32) * 1024 + prow * 64) + lcol;
  auto tr2 = [&](uint32_t a0, uint32_t a1) __attribute__((always_inline)) {
    union {
      p3_s16x4 h[2];
      bf16x8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a1);
    return u.v;
  };
  // clear reduction element q (a pixel of the step, wave-uniform) of an A fragment: lanes of half q >> 3, register (q & 7) >> 1, half-word q & 1
  auto clear_pixel = [&](bf16x8_t v, int q) __attribute__((always_inline)) {
    u32x4_t w = __builtin_bit_cast(u32x4_t, v);
    const uint32_t keep = (q & 1) ? 0x0000ffffu : 0xffff0000u;
    const uint32_t m = (lh == (q >> 3)) ? keep : 0xffffffffu;
    const int rq = (q & 7) >> 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] &= (r == rq) ? m : 0xffffffffu;
    return __builtin_bit_cast(bf16x8_t, w);
  };
  auto compute = [&](const int st, int xr) __attribute__((always_inline)) {  // xr = x coordinate of the step's first pixel
    const uint32_t so = (uint32_t)(st * STAGE);
    const int ql = xr == 0 ? 0 : a.W - xr;   // the pixel with x = 0 (16 or more: none in this step)
    const int qr = a.W - 1 - xr;             // the pixel with x = W - 1
    bf16x8_t pb[TN][3];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const uint32_t ad = offB + so + (uint32_t)((pc * GB + j) * 1024);
        pb[j][pc] = tr2(ad, ad + 256u);
      }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      bf16x8_t pa[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const uint32_t im = so + (uint32_t)(pc * GA * 1024);
        const uint32_t a0 = kx == 0 ? offA0[pc] + so : offA[kx][0] + im;
        const uint32_t a1 = kx == 2 ? offA2[pc] + so : offA[kx][1] + im;
        pa[pc] = tr2(a0, a1);
      }
      if (kx == 0 && ql < 16) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) pa[pc] = clear_pixel(pa[pc], ql);
      }
      if (kx == 2 && qr < 16) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) pa[pc] = clear_pixel(pa[pc], qr);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[kx][j] = mfma_x6(pa, pb[j], acc[kx][j]);
    }
  };
  const int nt = (k_end - k_begin + BK - 1) / BK;
  if (nt > 0) issue(0);
  int xr = __builtin_amdgcn_readfirstlane(k_begin - (int)fdiv((uint32_t)k_begin, a.dW) * a.W);  // k_begin mod W (every pixel row has W pixels)
  f32x16 acc2[FL ? 3 : 1][FL ? TN : 1];
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  }
  auto step = [&](const int st, int t) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < nt) {
      advance();
      issue(st ^ 1);
    }
    compute(st, xr);
    xr += BK;
    if (xr >= a.W) xr -= a.W;
    if constexpr (FL) {
      if ((t & 31) == 31) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] += acc[i][j][r], acc[i][j][r] = 0.f;
      }
    }
  };
  for (int t = 0; t < nt; t += 2) {
    step(0, t);
    if (t + 1 < nt) step(1, t + 1);
  }
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
  }
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (ky * 3 + kx) * a.C + cb + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + j * 32 + l31;
        if (col < a.Kout) atomicAdd(a.dwf + (int64_t)row * a.Kout + col, acc[kx][j][r]);
      }
    }
  }
}

// C % 64: with one 32-channel group per tile (GA = 1) the three taps share too little -- measured 47 against 72 TFLOP/s of the kernel above
static bool wgrad3x3_p3_ok(const fmi_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dil <= 1 && d->W >= 16 && d->OH == d->H && d->OW == d->W && d->C % 64 == 0;
}

// x3 / dy3: piece images of x [N][H][W][C] and dy [N][OH][OW][K] (dense tensors)
static int launch_wgrad_p3(const fmi_conv_desc* d, const uint16_t* x3, const uint16_t* dy3, float* dwf, hipStream_t st) {
  WgP3Args a{};
  a.x3 = x3; a.dy3 = dy3; a.dwf = dwf;
  ConvGeom& g = a.g;
  const int dl = d->dil > 1 ? d->dil : 1;
  g.N = d->N; g.IH = d->H; g.IW = d->W; g.C = d->C; g.cstride = d->C;
  g.GH = d->OH; g.GW = d->OW; g.S = d->stride;
  g.nty = d->kh; g.ntx = d->kw; g.dy0 = -d->pad; g.dx0 = -d->pad; g.ystep = dl; g.xstep = dl;
  g.kh0 = 0; g.kw0 = 0; g.khstep = 1; g.kwstep = 1; g.kw = d->kw;
  g.dGW = make_fastdiv(g.GW); g.dG = make_fastdiv(g.GH * g.GW); g.dC = make_fastdiv(g.C); g.dntx = make_fastdiv(g.ntx);
  a.Kout = d->K; a.Mrows = d->kh * d->kw * d->C; a.P = d->N * d->OH * d->OW;
  static const int tile_dbg = getenv("FMI_WG3_TILE") ? atoi(getenv("FMI_WG3_TILE")) : 0;  // experiment: 1 = the 8-wave 256 x 128 tile, 2 = never the tap-reuse kernel
  if (tile_dbg != 2 && wgrad3x3_p3_ok(d)) {
    Wg3x3Args w{};
    w.x3 = x3; w.dy3 = dy3; w.dwf = dwf;
    w.H = d->H; w.W = d->W; w.C = d->C; w.Kout = d->K; w.P = d->N * d->H * d->W;
    w.dW = make_fastdiv(d->W); w.dHW = make_fastdiv(d->H * d->W);
    const int bn3 = d->K <= 64 ? 64 : 128;
    const int64_t tm3 = 3 * (d->C / 64), tn3 = ceil_div64(d->K, bn3);
    int64_t ks = 2048 / (tm3 * tn3);
    const int64_t kmax3 = w.P / 512;
    if (ks > kmax3) ks = kmax3;
    if (ks < 1) ks = 1;
    if (ks >= 6) ks = (ks + 4) / 8 * 8;
    if (fmi_det()) ks = 1;
    w.kchunk = (int)(ceil_div64(ceil_div64(w.P, ks), 16) * 16);
    ks = ceil_div64(w.P, w.kchunk);
    w.tiles = (int)(tm3 * tn3);
    w.ksplit = (int)ks;
    w.xcd_splits = ks >= 8 ? 1 : 0;
    const int64_t nwg3 = w.xcd_splits ? tm3 * tn3 * ceil_div64(ks, 8) * 8 : tm3 * tn3;
    if (nwg3 > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
    const dim3 grid3((unsigned)nwg3, w.xcd_splits ? 1u : (unsigned)ks);
    const bool fl3 = w.kchunk > 640 && (fmi_det() || fmi_blocked_acc());
#define WG3X3_LAUNCH(GA_, TN_)                                                                                 \
  do {                                                                                                         \
    if (fl3) hipLaunchKernelGGL((wgrad3x3_p3_kernel<GA_, TN_, true>), grid3, dim3(256), 0, st, w, (int)tn3);   \
    else hipLaunchKernelGGL((wgrad3x3_p3_kernel<GA_, TN_>), grid3, dim3(256), 0, st, w, (int)tn3);             \
  } while (0)
    if (bn3 == 128) WG3X3_LAUNCH(2, 2);
    else WG3X3_LAUNCH(2, 1);
#undef WG3X3_LAUNCH
    return fmi_launch_status();
  }
  const bool big = tile_dbg == 1 && d->K > 64;
  const int bm = big ? 256 : 128;
  const int bn = d->K <= 32 ? 32 : (d->K <= 64 ? 64 : 128);
  const int64_t tm = ceil_div64(a.Mrows, bm), tn = ceil_div64(d->K, bn);
  int64_t ksplit = 2048 / (tm * tn);
  const int64_t kmax = a.P / 512;
  if (ksplit > kmax) ksplit = kmax;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535) ksplit = 65535;
  if (ksplit >= 6) ksplit = (ksplit + 4) / 8 * 8;  // splits are dealt to the 8 XCDs in groups of 8: keep the groups full
  if (fmi_det()) ksplit = 1;                       // reproducible mode
  a.kchunk = (int)(ceil_div64(ceil_div64(a.P, ksplit), 16) * 16);
  ksplit = ceil_div64(a.P, a.kchunk);
  a.tiles = (int)(tm * tn);
  a.ksplit = (int)ksplit;
  a.xcd_splits = ksplit >= 8 ? 1 : 0;
  const int64_t nwg = a.xcd_splits ? tm * tn * ceil_div64(ksplit, 8) * 8 : tm * tn;
  if (nwg > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)nwg, a.xcd_splits ? 1u : (unsigned)ksplit);
  const bool fl = a.kchunk > 640 && (fmi_det() || fmi_blocked_acc());  // blocked accumulation of a long unsplit pixel reduction
#define WG3_LAUNCH(TILE, NT)                                                                              \
  do {                                                                                                    \
    if (fl) hipLaunchKernelGGL((wgrad_p3_kernel<TILE, true>), grid, dim3(NT), 0, st, a, (int)tn);         \
    else hipLaunchKernelGGL((wgrad_p3_kernel<TILE>), grid, dim3(NT), 0, st, a, (int)tn);                  \
  } while (0)
  if (big) WG3_LAUNCH(WG3_256x128, 512);
  else if (bn == 32) WG3_LAUNCH(WG3_128x32, 256);
  else if (bn == 64) WG3_LAUNCH(WG3_128x64, 256);
  else WG3_LAUNCH(WG3_128x128, 256);
#undef WG3_LAUNCH
  return fmi_launch_status();
}
static bool wgrad_p3_ok(const fmi_conv_desc* d, const void* x3, const void* dy3) {
  return x3 && dy3 && d->C % 32 == 0 && d->K % 16 == 0 && d->x_cstride == d->C && d->y_cstride == d->K && d->pad_mode == 0 && ((uintptr_t)x3 & 15) == 0 &&
         ((uintptr_t)dy3 & 15) == 0 && (int64_t)d->N * d->H * d->W * 3 * d->C < (1ll << 40) && (int64_t)d->N * d->OH * d->OW < 0x7fffffffLL;
}
#endif
