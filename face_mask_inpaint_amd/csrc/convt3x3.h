// ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) forward with narrow outputs on large maps -- the upsampling convolutions of
// the generator's last ResBlockDecoders (base_function.py:297-305: 64 -> 32 and 32 -> 32 at 512^2 -> 1024^2, 128 -> 64 / 64 -> 64 one level
// down) -- as ONE launch with TAP REUSE, all four sub-pixel phases per workgroup.
//
// As four launches of the generic implicit GEMM (one per phase) a 128 x 32 tile stages 11 KB per 16-deep step for 192 matrix cycles per
// SIMD: 57 B/clk/CU against the 24 - 28 the L2 -> LDS path delivers, every phase streams the whole input again, and each tile's epilogue
// follows a reduction of only 4 - 16 steps: 47 - 81 TFLOP/s, 3 x the HBM floor.  (Measured and dropped on the way here: the same four GEMMs
// as one launch with the phases adjacent on an XCD -- slower, 1.29 vs 0.97 ms: the input re-reads were not the bound; a register-direct
// kernel without LDS -- slower, its weight fragments come from L2 per wave.)
//
// y[2i + a][2j + b] = sum over the taps (ky, kx) with ky = a + 1 (mod 2) ..., i.e.
//     a = 0: ky = 1 from input row i;            a = 1: ky = 0 from row i + 1, ky = 2 from row i;   the same in x.
// A step is one (16-channel group, input row offset dy in {0, 1}): the A image holds the tile's BM + 1 consecutive input pixels (flattened
// [N][H][W] order) at image row y + dy; the fragment of view dx in {0, 1} is rows l + dx (the pixel after the last one of an image row is
// cleared in registers).  dy = 0 serves six taps, dy = 1 three; the weight tiles ([tap][3 pieces][2 channel groups][32] 16-byte chunks,
// ready B fragments from the piece images of the adjoint's pack) ride along; four accumulators per wave, one per phase.  Per 16 channels:
// 16.5 KB of activations + 27 KB of weights for 54 x 32 matrix cycles per SIMD = 25 B/clk/CU.
#pragma once
#include "gemm_core.h"

#ifndef FMI_HOST_EMU
struct CT3Args {
  const float* x;       // convT input [N][H][W], pixel pitch cs, Cred channels
  const uint16_t* w3;   // pieces of the adjoint's pack: [3][9][Cred / 8][Nout][8]
  // optional SECOND ConvTranspose2d of the same geometry whose result is added (ResBlockDecoder: main path + bypass, base_function.py:297-305):
  // the reduction simply continues over its input's channels -- one output write, no residual round trip (x2 = nullptr: none)
  const float* x2;
  const uint16_t* w3b;
  int Cred2, cs2;
  int N, H, W, Cred, cs, Nout;
  FastDiv dW, dHW;
  ConvEp ep;            // rows = input pixels; phase (a, b) writes at (2 y + a, 2 x + b): py / px are set per phase in the kernel
};

// weight tiles of a step are staged in "slot" order: dy = 0: (ky, kx) = (1,1) (1,0) (1,2) (2,1) (2,0) (2,2);  dy = 1: (0,1) (0,0) (0,2)
__device__ __forceinline__ int ct3_tap(int dy, int slot) {
  const int ky = dy ? 0 : 1 + (slot >= 3), s3 = slot >= 3 ? slot - 3 : slot;
  return ky * 3 + (s3 == 0 ? 1 : (s3 == 1 ? 0 : 2));
}

template <int TN>
__global__ void __launch_bounds__(256, TN == 1 ? 3 : 2) convt3x3s2_dma_kernel(CT3Args a, int M, int tiles_n) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = 128, BN = 32 * TN, BK = 16;
  constexpr int RA = ((BM + 1 + 15) / 16) * 16;  // 144 rows
  constexpr int NIA = RA / 16;                   // 9 wave instructions of an A image
  constexpr int TAPCH = 3 * 2 * BN;              // 16-byte chunks of one weight tile
  constexpr int NIB = 6 * TAPCH / 64;            // wave instructions of six weight tiles
  constexpr int NLA = (NIA + 3) / 4, NLB = (NIB + 3) / 4;
  // stage 0 always holds a dy = 0 step (six weight tiles), stage 1 a dy = 1 step (three): 27 + 18 KB at 32 columns = three workgroups per CU
  constexpr int ABYTES = RA * BK * 4, STAGE = ABYTES + 6 * TAPCH * 16, STAGE1 = ABYTES + 3 * TAPCH * 16;
  static_assert(STAGE + STAGE1 <= 160 * 1024, "two stages in LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[STAGE + STAGE1];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = wid * 32;
  const int HW = a.H * a.W;
  const int ncg1 = a.Cred >> 4, ncg = ncg1 + (a.x2 ? a.Cred2 >> 4 : 0);
  const int nit = 2 * ncg;

  // A copy slots: wave instruction j * 4 + wid covers LDS rows 16 (j * 4 + wid) .. + 15, four lanes (k-quarters) per row; row r <-> input pixel m0 + r
  int64_t abase[NLA], abase2[NLA];
  int ay[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int r = (j * 4 + wid) * 16 + (lane >> 2);
    const int kq = ((lane & 3) ^ ((r >> 2) & 3)) * 4;
    const int an = m0 + r;
    ay[j] = 0x20000000;
    abase[j] = abase2[j] = 0;
    if (an < M && r < BM + 1) {
      const uint32_t n = fdiv((uint32_t)an, a.dHW);
      const uint32_t rem = (uint32_t)an - n * (uint32_t)HW;
      ay[j] = (int)fdiv(rem, a.dW);
      abase[j] = (int64_t)an * a.cs + kq;
      abase2[j] = (int64_t)an * a.cs2 + kq;
    }
  }
  // B copy slots: chunk p of [6 tap slots][3 pieces][2 channel groups][BN]
  int boff[NLB], bslot[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = (j * 4 + wid) * 64 + lane;
    const int slot = p / TAPCH, q = p - slot * TAPCH;
    const int piece = q / (2 * BN), r = q - piece * (2 * BN), kg = r / BN, n = r - kg * BN;
    bslot[j] = slot | (piece << 3);
    boff[j] = (n0 + n < a.Nout && slot < 6) ? (kg * a.Nout + n0 + n) * 8 : -1;
  }
  const int na_w = (NIA - wid + 3) / 4, nb_w = (NIB - wid + 3) / 4;
  // this lane's fragment row: is it the last pixel of an image row (view dx = 1 leaves the image)?
  bool xr;
  {
    const uint32_t an = (uint32_t)(m0 + wm + l31);
    const uint32_t rem = an - fdiv(an, a.dHW) * (uint32_t)HW;
    xr = rem - fdiv(rem, a.dW) * (uint32_t)a.W == (uint32_t)a.W - 1;
  }

  f32x16 acc[4][1][TN];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ph][0][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto glds16 = [&](const void* g, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
  };
  auto issue = [&](int it, int st) __attribute__((always_inline)) {
    const int cg_all = it >> 1, dy = it & 1;  // wave-uniform: channel group outer, row offset inner
    const bool second = cg_all >= ncg1;       // the second input's channels follow the first's
    const int c0 = (second ? cg_all - ncg1 : cg_all) * 16;
    const float* xs = second ? a.x2 : a.x;
    const uint16_t* ws = second ? a.w3b : a.w3;
    const int cred = second ? a.Cred2 : a.Cred;
    const int64_t wpiece = (int64_t)9 * cred * a.Nout;
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE) + (uint32_t)wid * 1024u);
    const uint32_t sb = sa + ABYTES;
    const int64_t aoff = (int64_t)dy * a.W * (second ? a.cs2 : a.cs) + c0;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (j >= na_w) break;
      const bool ok = ay[j] + dy < a.H;
      const void* g = ok ? (const void*)(xs + (second ? abase2[j] : abase[j]) + aoff) : (const void*)zchunk;
      glds16(g, sa + (uint32_t)(j * 4096));
    }
    const int ntap = dy ? 3 : 6;
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (j >= nb_w) break;
      const int slot = bslot[j] & 7;
      if ((j * 4 + wid) * 64 >= ntap * TAPCH) break;  // wave-uniform: dy = 1 stages three tiles only
      const int tap = ct3_tap(dy, slot);
      const void* g = (boff[j] >= 0 && slot < ntap)
                          ? (const void*)(ws + (int64_t)(bslot[j] >> 3) * wpiece + (((int64_t)tap * cred + c0) >> 3) * a.Nout * 8 + boff[j])
                          : (const void*)zchunk;
      glds16(g, sb + (uint32_t)(j * 4096));
    }
  };
  auto compute = [&](int st, const int dy) __attribute__((always_inline)) {
    const unsigned char* sa = lds + st * STAGE;
    const unsigned char* sb = sa + ABYTES;
    bf16x8_t pa[2][3];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int r = wm + l31 + dx, sw = (r >> 2) & 3;
      float4 v0 = *reinterpret_cast<const float4*>(sa + r * 64 + ((2 * lh) ^ sw) * 16);
      float4 v1 = *reinterpret_cast<const float4*>(sa + r * 64 + ((2 * lh + 1) ^ sw) * 16);
      if (dx == 1 && xr) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      split3_bf16(f, pa[dx]);
    }
    auto wfrag = [&](int slot, int j, bf16x8_t (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        b[pc] = *reinterpret_cast<const bf16x8_t*>(sb + ((slot * 3 + pc) * 2 * BN + lh * BN + j * 32 + l31) * 16);
    };
    // (slot, view, phase): dy = 0: (0, 0, 0) (1, 1, 1) (2, 0, 1) (3, 0, 2) (4, 1, 3) (5, 0, 3);  dy = 1: (0, 0, 2) (1, 1, 3) (2, 0, 3)
#pragma unroll
    for (int slot = 0; slot < (dy ? 3 : 6); ++slot) {
      const int view = (slot == 1 || slot == 4) ? 1 : 0;
      const int ph = dy ? (slot == 0 ? 2 : 3) : (slot == 0 ? 0 : (slot <= 2 ? 1 : (slot == 3 ? 2 : 3)));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8_t pb[3];
        wfrag(slot, j, pb);
        acc[ph][0][j] = mfma_x6(pa[view], pb, acc[ph][0][j]);
      }
    }
  };

  issue(0, 0);
  for (int it = 0; it < nit; it += 2) {  // unrolled over dy so that the tap table is static
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(it + 1, 1);
    compute(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 2 < nit) issue(it + 2, 0);
    compute(1, 1);
  }
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    ConvEp e = a.ep;  // by value: a reference into the kernel arguments would send the whole argument struct to scratch
    e.py = ph >> 1;
    e.px = ph & 1;
    store_tile<ConvEp, TileCfg<4, 1, 1, TN>>(e, acc[ph], M, a.Nout, m0 + wm, n0, lh, l31);
  }
}

// d: the descriptor of the convolution whose adjoint this is (fmi_conv2d_dgrad_f32): d->H x d->W = the ConvTranspose's OUTPUT
static bool convt3x3_eligible(const fmi_conv_desc* d, const float* dy, const uint16_t* w3) {
  return w3 && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad == 1 && d->dil <= 1 && d->H == 2 * d->OH && d->W == 2 * d->OW &&
         (d->K & 15) == 0 && d->C <= 64 && (d->C & 3) == 0 && (d->y_cstride & 3) == 0 && ((uintptr_t)dy & 15) == 0 &&
         (int64_t)d->N * d->OH * d->OW >= 32768 && (int64_t)d->N * d->OH * d->OW < (1ll << 30);
}
static int launch_convt3x3(CT3Args& a, int M, hipStream_t st) {
  const int64_t tm = ceil_div64(M, 128), tn = ceil_div64(a.Nout, 32);
  if (tm * tn > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((convt3x3s2_dma_kernel<1>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, a, M, (int)tn);
  return fmi_launch_status();
}
#endif
