// Experiment (round 2), measured SLOWER than conv3x3_dma_kernel<Tile128x64w, true> (156 vs 166 TFLOP/s at 8 x 128^2 256 -> 256, 148 vs 152 at
// 24 x 224^2 64 -> 64): kept for reference, not built.  Goes into csrc/conv3x3.h in front of conv3x3_eligible().
// =====================================================================================================
// Tap reuse with the activation image split ONCE per workgroup (x6.h): conv3x3_dma_kernel<T, true> still cuts every activation fragment
// into its bf16 pieces in the consuming wave -- three times per stage (once per kx tap, the same pixels shifted by one row) in each of
// the waves that share the rows.  Here the fp32 image of a stage ((BM + 2) pixels x 16 channels, LDS-DMA as before) is converted by
// all 256 threads into piece images in fragment layout, [piece][channel group][row] 16-byte chunks, while the MFMAs of the previous
// stage run; the taps then read ready fragments (one ds_read_b128 per piece, rows shifted by kx) and the weights arrive as piece
// images anyway (W3), so the main loop has no split arithmetic left (8 elements per thread and stage in the convert step).
//   iteration t:  wait own copies | barrier | copy A(t+2) -> SA[t & 1], W(t+1) -> SB[(t+1) & 1] | convert A(t+1): SA[(t+1) & 1] -> PA[(t+1) & 1]  ||  MFMAs of t: PA[t & 1], SB[t & 1]
// LDS per stage: 9 216 (fp32 image) + 12 480 (pieces) + 288 BN (weight pieces): 80 256 bytes at BN = 64, two workgroups per CU.
// =====================================================================================================
template <class T>
__global__ void __launch_bounds__(256) conv3x3_cvt_kernel(C3Args a, ConvEp ep, int M, int tiles_n, int ksplit, int it_chunk) {
  constexpr int BM = T::BM, BN = T::BN, BK = 16;
  constexpr int RA = ((BM + 2 + 15) / 16) * 16;  // rows of the fp32 image (multiple of the 16 rows one wave instruction writes)
  constexpr int RP = BM + 2;                     // rows of the piece images
  constexpr int NIA = RA / 16, NIB = 9 * BN / 32;
  constexpr int NLA = (NIA + 3) / 4, NLB = (NIB + 3) / 4;
  constexpr int SA_B = RA * BK * 4, PA_B = 3 * 2 * RP * 16, SB_B = 3 * 3 * 2 * BN * 16;  // bytes
  static_assert(FMI_X6, "piece images feed the bf16 products");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_c3[];
  unsigned char* SA = lds_c3;                // [2][SA_B]
  unsigned char* SB = SA + 2 * SA_B;         // [2][SB_B]
  unsigned char* PA = SB + 2 * SB_B;         // [2][PA_B]

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
  int boff[NLB], bkx[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {  // chunk p of [3 kx][3 pieces][2 channel groups][BN]
    const int p = (j * 4 + wid) * 64 + lane;
    const int per_tap = 3 * 2 * BN;
    const int kx = p / per_tap, q = p - kx * per_tap;
    const int piece = q / (2 * BN), r = q - piece * (2 * BN), kg = r / BN, n = r - kg * BN;
    bkx[j] = kx | (piece << 2);
    boff[j] = (n0 + n < a.Nout && kx < 3) ? (kg * a.Nout + n0 + n) * 8 : -1;
  }
  const int na_w = (NIA - wid + 3) / 4, nb_w = (NIB - wid + 3) / 4;
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

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_c3;
  auto glds16 = [&](const void* g, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
  };
  auto issue_a = [&](int it, int st) {
    const int ky = it / cchunks, c0 = (it - ky * cchunks) * 16;  // wave-uniform
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * SA_B) + (uint32_t)wid * 1024u);
    const int64_t aoff = (int64_t)(ky - 1) * a.W * a.cs + c0;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (j >= na_w) break;
      const bool ok = (unsigned)(ay[j] + ky - 1) < (unsigned)a.H;
      const float* g = ok ? a.x + abase[j] + aoff : fmi_chunk_zero;
      glds16(g, sa + (uint32_t)(j * 4096));
    }
  };
  auto issue_b = [&](int it, int st) {
    const int ky = it / cchunks, c0 = (it - ky * cchunks) * 16;
    const uint32_t sb = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(2 * SA_B + st * SB_B) + (uint32_t)wid * 1024u);
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (j >= nb_w) break;
      const int tap = ky * 3 + (bkx[j] & 3);
      const void* g = boff[j] >= 0 ? (const void*)(a.w3 + (int64_t)(bkx[j] >> 2) * 9 * a.C * a.Nout + (((int64_t)(a.flip ? 8 - tap : tap) * a.C + c0) >> 3) * a.Nout * 8 + boff[j])
                                   : (const void*)fmi_chunk_zero;
      glds16(g, sb + (uint32_t)(j * 4096));
    }
  };
  auto convert_item = [&](const float* s, unsigned char* p, int item) {  // item = (channel group, row) of the piece images
    const int kg = item / RP, r = item - kg * RP, sw = (r >> 2) & 3;
    const float4 v0 = *reinterpret_cast<const float4*>(s + r * 16 + ((2 * kg) ^ sw) * 4);
    const float4 v1 = *reinterpret_cast<const float4*>(s + r * 16 + ((2 * kg + 1) ^ sw) * 4);
    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    bf16x8_t q[3];
    split3_bf16(f, q);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<bf16x8_t*>(p + ((pc * 2 + kg) * RP + r) * 16) = q[pc];
  };
  auto convert = [&](int st) {
    const float* s = reinterpret_cast<const float*>(SA + st * SA_B);
    unsigned char* p = PA + st * PA_B;
    convert_item(s, p, tid);
    if (tid < 2 * RP - 256) convert_item(s, p, 256 + tid);
  };
  auto compute = [&](int st) {
    const unsigned char* pa = PA + st * PA_B;
    const unsigned char* sb = SB + st * SB_B;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      bf16x8_t fa[T::TM][3], fb[T::TN][3];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const bool zero = (kx == 0 && xl[i]) || (kx == 2 && xr[i]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          u32x4_t v = *reinterpret_cast<const u32x4_t*>(pa + ((pc * 2 + lh) * RP + wm + i * 32 + l31 + kx) * 16);
          if (zero) v = u32x4_t{0u, 0u, 0u, 0u};
          fa[i][pc] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          fb[j][pc] = *reinterpret_cast<const bf16x8_t*>(sb + ((kx * 3 + pc) * 2 * BN + lh * BN + wn + j * 32 + l31) * 16);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(fa[i], fb[j], acc[i][j]);
    }
  };

  const int nt = it_end - it_begin;
  if (nt > 0) {
    issue_a(it_begin, 0);
    issue_b(it_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nt > 1) issue_a(it_begin + 1, 1);
    convert(0);
  }
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of A(t+1) and W(t) have landed
    __syncthreads();                                   // ... everyone's; the pieces of A(t) are complete; PA / SB of stage (t+1)&1 and SA of stage t&1 are free
    if (t + 2 < nt) issue_a(it_begin + t + 2, t & 1);
    if (t + 1 < nt) {
      issue_b(it_begin + t + 1, (t + 1) & 1);
      convert((t + 1) & 1);
    }
    compute(t & 1);
  }
  store_tile<ConvEp, T>(ep, acc, M, a.Nout, m0 + wm, n0 + wn, lh, l31);
}

