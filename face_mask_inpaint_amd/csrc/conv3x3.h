// 3x3 stride-1 zero-padded convolution (forward and adjoint) on the LDS-DMA pipeline with TAP REUSE along x.
//
// The generic implicit GEMM (gemm_core.h) fetches one 16-channel operand tile per tap: the three taps of a kernel row read
// the SAME pixels shifted by one, so 2/3 of that traffic (and 2/3 of the barriers) are redundant.  Here a k-step is one
// (kernel row ky, 16-channel chunk): the A image holds BM + 2 consecutive pixels of the flattened [N][H][W] order, taken from
// image row y + ky - 1, and the three kx taps read fragment rows shifted by 0 / 1 / 2; three weight tiles ride along.  Per
// barrier a wave runs 3 x 8 x TM x TN MFMAs instead of 8 x TM x TN.
//   LDS row r  <->  anchor m0 - 1 + r (pixel (n, y, x) of the flattened order) read at image row y + ky - 1; rows outside the
//                   image are zero chunks.  Anchor i, tap kx reads LDS row i + kx = pixel x + kx - 1; where that wrapped
//                   around a row end (x = 0 with kx = 0, x = W-1 with kx = 2) the fragment is zeroed in registers.
//   adjoint    :    dx = conv(dy, flipped taps) with reduction over the conv's output channels: same kernel, tap index 8 - t.
#pragma once
#include "gemm_core.h"

#ifndef FMI_HOST_EMU
struct C3Args {
  const float* x;  // gathered image [N][H][W] pixel pitch cs, C reduction channels used
  const float* w;  // packed weights [9][C][Nout]
  int N, H, W, C, cs, Nout, flip;
  FastDiv dW, dHW;
  const uint16_t* w3;  // W3 kernels: the same weights as bf16 pieces [3][9][C / 8][Nout][8] (ConvWX3 in gemm_core.h)
};

// W3: the weight tiles arrive as bf16 piece images ([3 kx][3 pieces][2 channel groups][BN] 16-byte chunks per stage) and are read as
// ready B fragments; only the activation fragments are split in registers.
// FL: blocked accumulation (a second accumulator set, flushed every 10 steps = 480 reduction elements), see conv_p3.h
template <class T, bool W3, bool FL = false>
__global__ void __launch_bounds__(256) conv3x3_dma_kernel(C3Args a, ConvEp ep, int M, int tiles_n, int ksplit, int it_chunk) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, BK = 16;
  constexpr int RA = ((BM + 2 + 15) / 16) * 16;  // rows of the A image (multiple of the 16 rows one wave instruction writes)
  constexpr int NIA = RA / 16;                   // wave instructions of an A image
  constexpr int NIB = W3 ? 9 * BN / 32 : 3 * BN / 16;  // wave instructions of the three B tiles ([3][16 k][BN] floats, or [3][3][2][BN] chunks)
  constexpr int NLA = (NIA + 3) / 4, NLB = (NIB + 3) / 4;
  constexpr int STAGE = RA * BK + (W3 ? 3 * 3 * BN * 8 : 3 * BK * BN);  // floats
  static_assert(!W3 || FMI_X6, "piece images feed the bf16 products");
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int zs = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int cchunks = a.C >> 4, nit_all = 3 * cchunks;
  const int it_begin = zs * it_chunk;
  int it_end = it_begin + it_chunk;
  if (it_end > nit_all) it_end = nit_all;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;
  const int HW = a.H * a.W;

  // A copy slots of this thread: wave instruction j*4 + wid covers LDS rows 16*(j*4+wid) .. +15, 4 lanes (k-quarters) per row
  int64_t abase[NLA];
  int ay[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int r = (j * 4 + wid) * 16 + (lane >> 2);
    const int kq = ((lane & 3) ^ ((r >> 2) & 3)) * 4;
    const int an = m0 - 1 + r;
    ay[j] = -0x20000000;
    abase[j] = 0;
    if (an >= 0 && an < M && r < BM + 2) {
      const uint32_t n = fdiv((uint32_t)an, a.dHW);
      const uint32_t rem = (uint32_t)an - n * (uint32_t)HW;
      ay[j] = (int)fdiv(rem, a.dW);
      abase[j] = (int64_t)an * a.cs + kq;
    }
  }
  // B copy slots: chunk p of [3][16][BN/4]
  int boff[NLB], bkx[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = (j * 4 + wid) * 64 + lane;
    if (W3) {  // chunk p of [3 kx][3 pieces][2 channel groups][BN]; boff = element offset inside a piece image, piece in the high bits of bkx
      const int per_tap = 3 * 2 * BN;
      const int kx = p / per_tap, q = p - kx * per_tap;
      const int piece = q / (2 * BN), r = q - piece * (2 * BN), kg = r / BN, n = r - kg * BN;
      bkx[j] = kx | (piece << 2);
      boff[j] = (n0 + n < a.Nout && kx < 3) ? (kg * a.Nout + n0 + n) * 8 : -1;
    } else {
      const int per_tap = 16 * (BN / 4);
      const int kx = p / per_tap, q = p - kx * per_tap;
      const int k = q / (BN / 4), n = (q - k * (BN / 4)) * 4;
      bkx[j] = kx;
      boff[j] = (n0 + n < a.Nout && kx < 3) ? k * a.Nout + n0 + n : -1;
    }
  }
  const int na_w = (NIA - wid + 3) / 4, nb_w = (NIB - wid + 3) / 4;  // instructions wave `wid` issues per stage
  // x-wrap masks of this lane's fragment rows
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

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto glds16 = [&](const float* g, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
  };
  auto issue = [&](int it, int st) {
#ifdef FMI_C3_KY_OUTER
    const int ky = it / cchunks, c0 = (it - ky * cchunks) * 16;  // wave-uniform
#else
    // channel chunk OUTER, kernel row INNER: the three kernel rows re-read one channel chunk of neighbouring image rows while it is still in
    // the XCD's L2 (with the kernel row outer every activation byte came three times from beyond L2)
    const int cc_ = it / 3, ky = it - 3 * cc_, c0 = cc_ * 16;  // wave-uniform
#endif
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE) * 4u + (uint32_t)wid * 1024u);  // + this wave's slot
    const uint32_t sb = sa + RA * BK * 4;
    const int64_t aoff = (int64_t)(ky - 1) * a.W * a.cs + c0;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (j >= na_w) break;
      const bool ok = (unsigned)(ay[j] + ky - 1) < (unsigned)a.H;
      const float* g = ok ? a.x + abase[j] + aoff : zchunk;
      glds16(g, sa + (uint32_t)(j * 4096));
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (j >= nb_w) break;
      const int tap = ky * 3 + (bkx[j] & 3);
      const float* g;
      if (W3)
        g = boff[j] >= 0 ? (const float*)(a.w3 + (int64_t)(bkx[j] >> 2) * 9 * a.C * a.Nout + (((int64_t)(a.flip ? 8 - tap : tap) * a.C + c0) >> 3) * a.Nout * 8 + boff[j])
                         : zchunk;
      else
        g = boff[j] >= 0 ? a.w + ((int64_t)(a.flip ? 8 - tap : tap) * a.C + c0) * a.Nout + boff[j] : zchunk;
      glds16(g, sb + (uint32_t)(j * 4096));
    }
  };
  auto compute = [&](int st) {
    const float* sa = lds + st * STAGE;
    const float* sb = sa + RA * BK;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float fa[T::TM][8], fb[T::TN][8];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const int r = wm + i * 32 + l31 + kx, sw = (r >> 2) & 3;
        float4 v0 = *reinterpret_cast<const float4*>(sa + r * 16 + ((2 * lh) ^ sw) * 4);
        float4 v1 = *reinterpret_cast<const float4*>(sa + r * 16 + ((2 * lh + 1) ^ sw) * 4);
        if ((kx == 0 && xl[i]) || (kx == 2 && xr[i])) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        fa[i][0] = v0.x, fa[i][1] = v0.y, fa[i][2] = v0.z, fa[i][3] = v0.w;
        fa[i][4] = v1.x, fa[i][5] = v1.y, fa[i][6] = v1.z, fa[i][7] = v1.w;
      }
      if (!W3) {
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
          for (int s = 0; s < 8; ++s) fb[j][s] = sb[(kx * 16 + 8 * lh + s) * BN + wn + j * 32 + l31];
      }
#if FMI_X6
      bf16x8_t pa[T::TM][3], pb[T::TN][3];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) split3_bf16(fa[i], pa[i]);
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        if (W3) {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            pb[j][pc] = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const unsigned char*>(sb) + ((kx * 3 + pc) * 2 * BN + lh * BN + wn + j * 32 + l31) * 16);
        } else {
          split3_bf16(fb[j], pb[j]);
        }
      }
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(pa[i], pb[j], acc[i][j]);
#else
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
#endif
    }
  };

  if (it_begin < it_end) issue(it_begin, 0);
  int st = 0, since = 0;
  f32x16 acc2[FL ? T::TM : 1][FL ? T::TN : 1];
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  }
  for (int it = it_begin; it < it_end; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of step `it` have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 1 < it_end) issue(it + 1, st ^ 1);
    compute(st);
    st ^= 1;
    if constexpr (FL) {
      if (++since == 10 || it + 1 == it_end) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              acc2[i][j][r] += acc[i][j][r];
              acc[i][j][r] = 0.f;
            }
        since = 0;
      }
    }
  }
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = acc2[i][j];
  }
  store_tile<ConvEp, T>(ep, acc, M, a.Nout, m0 + wm, n0 + wn, lh, l31);
}

// geometry the tap-reuse kernel takes (the caller falls back to the generic path otherwise).  C >= 64 or a narrow output: with 32
// reduction channels a tile has only 6 k-steps and, for 64 outputs, the heavier prologue loses (52 -> 42 TFLOP/s at 8 x 128^2, 32 -> 64)
// while 32 -> 32 still gains (71 -> 81)
static bool conv3x3_eligible(const float* x, const float* w, int C, int cs, int Nout, int64_t pixels) {
  return (C & 15) == 0 && (C >= 64 || Nout <= 32) && (cs & 3) == 0 && (Nout & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && pixels < (1ll << 30);
}

using Tile128x64w = TileCfg<4, 1, 1, 2>;  // 128 x 64 with every wave spanning both column tiles: one activation split per 12 MFMAs

static int launch_conv3x3(const C3Args& a, const ConvEp& ep, int M, int ksplit, hipStream_t st) {
  const int nit = 3 * (a.C >> 4);
  if (ksplit > nit) ksplit = nit;
  const int it_chunk = (nit + ksplit - 1) / ksplit;
  ksplit = (nit + it_chunk - 1) / it_chunk;
#define C3_LAUNCH_(TILE, W3)                                                                                                  \
  do {                                                                                                                        \
    const int64_t tm = ceil_div64(M, TILE::BM), tn = ceil_div64(a.Nout, TILE::BN);                                            \
    if (fl)                                                                                                                   \
      hipLaunchKernelGGL((conv3x3_dma_kernel<TILE, W3, true>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(256), 0, st, a, ep, M, (int)tn, \
                         ksplit, it_chunk);                                                                                   \
    else                                                                                                                      \
      hipLaunchKernelGGL((conv3x3_dma_kernel<TILE, W3>), dim3((unsigned)(tm * tn), (unsigned)ksplit), dim3(256), 0, st, a, ep, M, (int)tn, ksplit, \
                         it_chunk);                                                                                           \
  } while (0)
#define C3_LAUNCH(TILE) C3_LAUNCH_(TILE, false)
  auto wgs = [&](int bm, int bn) { return ceil_div64(M, bm) * ceil_div64(a.Nout, bn) * ksplit; };
  const int N = a.Nout;
  const bool fl = it_chunk > 13 && (fmi_det() || fmi_blocked_acc());  // blocked accumulation of a long unsplit reduction (conv_p3.h)
#if FMI_X6
  static const bool w3_off = getenv("FMI_W3_OFF") != nullptr;  // debug A/B: split the weight fragments in registers as well
  // piece images of the weights: 2 x 27 KB of LDS at 64 columns (two workgroups per CU; a 128-column piece tile would leave one).  Measured
  // at 8 x 128^2 256 -> 256 / 24 x 224^2 64 -> 64 / 8 x 32^2 128 -> 128: 163 / 154 / 81 TFLOP/s against 170 / 144 / 75 with both operands split in the
  // consuming waves (128 x 128 tiles): taken for narrow outputs and for launches too small to fill the chip with the large tile
  if (a.w3 && N > 32 && !w3_off && (N <= 64 || wgs(128, 128) < 800)) {
    C3_LAUNCH_(Tile128x64w, true);
    return fmi_launch_status();
  }
#endif
  if (N <= 32) {
    C3_LAUNCH(Tile128x32);
  } else if (N <= 64) {
    if (M > 64 && wgs(128, 64) >= 384) C3_LAUNCH(Tile128x64);
    else C3_LAUNCH(Tile64x64);
  } else {
    // between one and two rounds of 128x128 workgroups (3 per CU) the fullest CUs set the time: halve the work unit instead
    // (VGG 28^2: 588 tiles = 2.3 per CU, 94 -> 100 TFLOP/s with 64x128; at 1176 tiles the 128x128 tile wins again, 114 vs 102)
    if (M > 64 && wgs(128, 128) >= 800) C3_LAUNCH(Tile128x128);
    else C3_LAUNCH(Tile64x128);
  }
#undef C3_LAUNCH
#undef C3_LAUNCH_
  return fmi_launch_status();
}
#endif
