// Implicit-GEMM core on the bf16 matrix pipe with the operand split done ONCE per workgroup ("convert pass").
//
// gemm_dma_f32_kernel (gemm_core.h) under FMI_X6 cuts every fp32 fragment into its three bf16 pieces in the wave that consumes it: a
// 128 x 128 tile is split twice over (two waves share each fragment) at 5.5 VALU instructions per element, and the kernel is VALU bound
// (161 TFLOP/s at 8 x 128^2 256 -> 256, 220 with the split arithmetic removed).  Here the LDS-DMA still brings fp32 tiles (same
// loaders, same images), but a tile is converted once, by all 256 threads, into piece images laid out as MFMA fragments:
//     P[piece][k-group (2)][row] : one 16-byte chunk = 8 consecutive reduction indices of one row / column
// and the main loop reads ready fragments (one conflict-free ds_read_b128 per piece).  Pipeline, one barrier per 16-deep tile t:
//     wait DMA(t+1) | barrier | issue DMA(t+2) -> staging[t & 1] | convert(t+1): staging[(t+1) & 1] -> P[(t+1) & 1]  ||  MFMAs of tile t from P[t & 1]
// (the convert of the next tile and the MFMAs of the current one are independent and interleave in one basic block).
// LDS: staging 2 x (BM + BN) x 64 B + pieces 2 x (BM + BN) x 96 B (128 x 128: 80 KB, two workgroups per CU).
#pragma once

#ifndef FMI_HOST_EMU
template <class LA, class LB, class EP, class T>
__global__ void __launch_bounds__(256) gemm_x6_kernel(LA la, LB lb, EP ep, int M, int N, int K, int tiles_n, int ksplit, int kchunk) {
  constexpr int BM = T::BM, BN = T::BN, BK = 16;
  constexpr int NLA = (BM + 63) / 64, NLB = (BN + 63) / 64;
  constexpr int STAGE = (BM + BN) * BK;        // floats per staging buffer
  constexpr int PIMG_A = 3 * 2 * BM * 16;      // bytes of the A piece images of one tile
  constexpr int PIMG = 3 * 2 * (BM + BN) * 16; // bytes of one piece buffer (A then B)
  extern __shared__ __attribute__((aligned(1024))) float lds_x6[];
  float* lds = lds_x6;                                                               // [2][STAGE]
  unsigned char* pcs = reinterpret_cast<unsigned char*>(lds_x6 + 2 * STAGE);         // [2][PIMG]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int zb = blockIdx.y / ksplit, zs = blockIdx.y - zb * ksplit;
  la.set_batch(zb);
  lb.set_batch(zb);
  ep.set_batch(zb);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_begin = zs * kchunk;
  int k_end = k_begin + kchunk;
  if (k_end > K) k_end = K;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;

  // copy slots (as in gemm_dma_f32_kernel): position p = j*256 + tid of the lane-linear image
  typename LA::DCtx da[NLA];
  typename LB::DCtx db[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int p = j * 256 + tid;
    int x, k;
    if (LA::KMODE) {
      x = p >> 2;
      k = ((p & 3) ^ ((x >> 2) & 3)) * 4;
    } else {
      k = p / (BM / 4);
      x = (p % (BM / 4)) * 4;
    }
    da[j] = la.dprep(m0 + x, k);
    la.dstart(da[j], k_begin);
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = j * 256 + tid;
    int x, k;
    if (LB::KMODE) {
      x = p >> 2;
      k = ((p & 3) ^ ((x >> 2) & 3)) * 4;
    } else {
      k = p / (BN / 4);
      x = (p % (BN / 4)) * 4;
    }
    db[j] = lb.dprep(n0 + x, k);
    lb.dstart(db[j], k_begin);
  }
  const int na_w = (BM % 64 == 0) ? NLA : (wid * 64 < BM * 4 ? 1 : 0);
  const int nb_w = (BN % 64 == 0) ? NLB : (wid * 64 < BN * 4 ? 1 : 0);

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
  auto issue = [&](int k0, int st) {
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE + wid * 256) * 4u);
    const uint32_t sb = sa + BM * BK * 4;
    const typename LA::Tile ta = la.tile(k0);
    const typename LB::Tile tb = lb.tile(k0);
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (BM % 64 != 0 && !na_w) break;
      const float* g = la.chunk(da[j], ta);
      if (!g) g = fmi_chunk_zero;
      glds16(g, sa + j * 4096);
      la.advance(da[j]);
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (BN % 64 != 0 && !nb_w) break;
      const float* g = lb.chunk(db[j], tb);
      if (!g) g = fmi_chunk_zero;
      glds16(g, sb + j * 4096);
      lb.advance(db[j]);
    }
  };
  // one operand tile: staged fp32 image -> three piece images [piece][k-group][row] of 16-byte chunks; item = (row, k-group)
  auto convert_one = [&](const float* s, unsigned char* p, const bool kmode, const int BX, const int item) {
    const int r = item % BX, kg = item / BX;
    float f[8];
    if (kmode) {
      const int sw = (r >> 2) & 3;
      const float4 v0 = *reinterpret_cast<const float4*>(s + r * 16 + ((2 * kg) ^ sw) * 4);
      const float4 v1 = *reinterpret_cast<const float4*>(s + r * 16 + ((2 * kg + 1) ^ sw) * 4);
      f[0] = v0.x, f[1] = v0.y, f[2] = v0.z, f[3] = v0.w, f[4] = v1.x, f[5] = v1.y, f[6] = v1.z, f[7] = v1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = s[(8 * kg + j) * BX + r];
    }
    bf16x8_t q[3];
    split3_bf16(f, q);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<bf16x8_t*>(p + ((pc * 2 + kg) * BX + r) * 16) = q[pc];
  };
  auto convert = [&](int st, int pb) {
    const float* sa = lds + st * STAGE;
    const float* sb = sa + BM * BK;
    unsigned char* pa = pcs + pb * PIMG;
    unsigned char* pbb = pa + PIMG_A;
#pragma unroll
    for (int it = 0; it < (2 * BM + 255) / 256; ++it) {
      const int item = it * 256 + tid;
      if (2 * BM % 256 == 0 || item < 2 * BM) convert_one(sa, pa, LA::KMODE, BM, item);
    }
#pragma unroll
    for (int it = 0; it < (2 * BN + 255) / 256; ++it) {
      const int item = it * 256 + tid;
      if (2 * BN % 256 == 0 || item < 2 * BN) convert_one(sb, pbb, LB::KMODE, BN, item);
    }
  };
  auto compute = [&](int pb) {
    const unsigned char* pa = pcs + pb * PIMG;
    const unsigned char* pbb = pa + PIMG_A;
    bf16x8_t fa[T::TM][3], fb[T::TN][3];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) fa[i][pc] = *reinterpret_cast<const bf16x8_t*>(pa + ((pc * 2 + lh) * BM + wm + i * 32 + l31) * 16);
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) fb[j][pc] = *reinterpret_cast<const bf16x8_t*>(pbb + ((pc * 2 + lh) * BN + wn + j * 32 + l31) * 16);
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(fa[i], fb[j], acc[i][j]);
  };

  const int nt = (k_end - k_begin + BK - 1) / BK;
  if (nt > 0) {
    issue(k_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nt > 1) issue(k_begin + BK, 1);
    convert(0, 0);
  }
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of tile t+1 have landed
    __syncthreads();                                   // ... everyone's; the piece images of tile t are complete; P[(t+1)&1] and staging[t&1] are free
    if (t + 2 < nt) issue(k_begin + (t + 2) * BK, t & 1);
    if (t + 1 < nt) convert((t + 1) & 1, (t + 1) & 1);
    compute(t & 1);
  }

  store_tile<EP, T>(ep, acc, M, N, m0 + wm, n0 + wn, lh, l31);
}
#endif
