// EXPERIMENT (round 3, measured, not built): the fused attention backward with TWO waves per key block.
//
// Idea: the single-wave kernel (attn_bwd2_x6_kernel) is bound by ~2 400 vector instructions around 264 MFMAs per
// query tile on one wave per SIMD.  Here waves w and w + 4 of a 512-thread workgroup share key block w & 3: each owns half the value
// channels (64 dV accumulator + 64 V registers) and one d-tile, so two waves fit per SIMD (256 registers each, ~20 spilled) and the vector
// work per wave drops to a third.  The pair exchanges one partial dP tile (4 KB) and the dS^T piece image through LDS.
//
// Result on 8 x T 16384 x d 64 x C 256 (MI355X): CORRECT (all attention tests incl. the float64 check and the full-size properties pass),
// and NOT faster: backward 18.1 - 18.7 ms against 17.8 - 18.2 ms for the single-wave kernel, with five workgroup barriers per tile and with
// the two intra-pair hand-offs as LDS flags alike.  In-kernel cycle stamps of one query tile (s_memtime; waves 0 / 4 of one workgroup):
//     S + partial dP of both waves          ~8 000 cycles   (144 MFMAs = 4 608 matrix cycles; V is still re-split every tile: 352 vector
//                                                            instructions per wave on the SIMD's one vector pipe)
//     dS (wave B) || dV (wave A)            ~3 100          (dS is 2 000 cycles of LDS latency + split arithmetic, nothing of it matrix work)
//     dV (wave B) || dK / dQ (wave A)       ~3 400          (transposed LDS reads with no register prefetch: 256 registers are taken)
//     dK / dQ (wave B), wave A idle         ~2 200
//     partial dQ tiles, staging of the next tile, cross-key-block sum, atomics, 3 barriers   ~2 700 with no matrix work at all
//   = ~19 500 cycles per tile against 9 216 matrix cycles (the single-wave kernel: ~17 400 against 8 448).
// What it shows: the dependency chain S -> dP -> dS -> {dK, dQ} leaves the second wave without matrix work exactly where the first does
// vector / LDS work, unless the V pieces stay resident (no room: 96 registers) -- splitting channels across waves moves the stall, it does
// not remove it.  The launcher fragment and the kernel are kept here as they were built against csrc/attention.hip of this round.
//
// ---- launcher fragment (inside fmi_attention_bwd_f32, second-structure branch) ----
#if 0
    static const bool bwd3 = !(getenv("FMI_ATT_BWD3") && getenv("FMI_ATT_BWD3")[0] == '0');
    if (FMI_X6 && bwd3 && D == 64 && nct == 8) {  // two waves per key block, channel / d halves (attn_bwd3_x6_kernel)
      constexpr int lds3 = 3 * 32 * 2 * 256 + 3 * 32 * 192 + 256 + 4 * 3 * 32 * (2 * 64 + 16) + 4 * 8192 + 64;
      static fmi_attr_flags attr_set3{};
      int attr_set3_dev;
      if (fmi_attr_needed(attr_set3, attr_set3_dev)) {
        if (hipFuncSetAttribute((const void*)attn_bwd3_x6_kernel<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3) != hipSuccess) return FMI_ERR_LAUNCH;
        fmi_attr_mark(attr_set3, attr_set3_dev);
      }
      for (int kb = 0; kb < (det ? (int)grid2.x : 1); ++kb)
        hipLaunchKernelGGL((attn_bwd3_x6_kernel<64, 8>), det ? dim3(1, grid2.y) : grid2, dim3(512), lds3, st, q, v1, v2, go1, go2, lse,
                           (const float*)delta_scratch, gv1, gv2, gq_zeroed, T, C1, C2, kb);
      return fmi_launch_status();
    }
#endif

// =====================================================================================================
// Backward, third structure (round 3): TWO waves per key block, each owning HALF the value channels and half of d.
//
// The single-wave form above is bound by the ~2 400 vector instructions around its 264 MFMAs per query tile on ONE wave per SIMD (the
// 128 dV + 32 dK accumulator and 128 V registers leave no room for a second wave; V is re-split into pieces every tile).  Here a
// workgroup is 8 waves: waves w and w + 4 (the same SIMD) share key block w & 3.  Wave "A" (w < 4) takes value channels [0, CT/2) and
// d-tile 0, wave "B" channels [CT/2, CT) and d-tile 1:
//   S, P              both (24 MFMAs, redundant: the pair needs P twice anyway)
//   dP = gO V^T       each over its channel half (partial sums); A hands its partial tile to B through LDS (4 KB) and moves on
//   dS = P (dP - d)   B only; written as the transposed piece image [key][q] that both waves read (as row fragments for dK,
//                     transposed for dQ).  While B does this vector work A already runs its dV MFMAs on the shared SIMD.
//   dV^T += gO^T P    each its channel half (48 MFMAs)
//   dK^T, dQ          each its d-tile (12 + 12 MFMAs)
// = 144 MFMAs per wave and tile (288 per pair against 264), 64 + 16 accumulator and 64 V registers per wave: two waves per SIMD fit,
// the vector work per wave drops to a third (V split halves, staging split shared by 8 waves, dS split on A only).
// LDS (155 904 B): gO pieces 48 KB, Q pieces 18 KB, four key-block piece images 54 KB, one 8 KB slot per pair that holds in turn B's
// partial dP, the dS^T piece image and the pair's two partial dQ tiles.  The two hand-offs inside a pair are LDS flags (publish / await);
// three workgroup barriers per query tile order the slot's reuse and the staging of the next tile.
// =====================================================================================================
#ifdef FMI_ATT_STAMP  // diagnostic build only (tools/bench_tools/build_flags.sh): cycle stamps of the phases of ONE query tile, waves 0 and 4 of workgroup (0, 0)
__device__ unsigned long long fmi_att_stamps[2][16];
extern "C" int fmi_debug_attn_stamps(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(fmi_att_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? FMI_OK : FMI_ERR_LAUNCH;
}
#define ATT_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && i0 == 32 * 100 && (wid & 3) == 0 && lane == 0)                        \
      fmi_att_stamps[half][i] = __builtin_amdgcn_s_memtime();                                                     \
  } while (0)
#else
#define ATT_STAMP(i)
#endif
template <int D, int NCT>
__global__ void __launch_bounds__(512, 1) attn_bwd3_x6_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                              const float* __restrict__ v2, const float* __restrict__ g1,
                                                              const float* __restrict__ g2, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, float* __restrict__ gv1,
                                                              float* __restrict__ gv2, float* __restrict__ gq, int T, int C1, int C2, int kb0) {
  static_assert(D == 64 && NCT % 2 == 0, "two d-tiles, an even number of channel tiles");
  constexpr int CT = NCT * 32, CH = NCT / 2, NDT = D / 32;
  constexpr int NQL = (8 * D) / 512 > 0 ? (8 * D) / 512 : 1, NVL = (8 * CT) / 512;
  constexpr int GP = 2 * CT, GIMG = 32 * GP;   // gO piece image: row pitch, bytes per piece
  constexpr int QP = 192, QIMG = 32 * QP;      // Q piece image
  constexpr int TIMG = 32 * 64;                // dS^T piece image: [32 keys][32 q] bf16
  constexpr int KP = 2 * D + 16, KIMG = 32 * KP;
  constexpr int PSLOT = 8192;                  // per pair: partial dP (4 KB) | dS^T pieces (6 KB) | two partial dQ tiles (2 x 4 KB)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Gs = smem_b;                                     // [3][32][GP]
  unsigned char* Qs = Gs + 3 * GIMG;                              // [3][32][QP]
  float* lse_i = reinterpret_cast<float*>(Qs + 3 * QIMG);         // [32]
  float* del_i = lse_i + 32;                                      // [32]
  unsigned char* Ks = reinterpret_cast<unsigned char*>(del_i + 32);  // [4][3][32][KP]
  unsigned char* Ps = Ks + 4 * 3 * KIMG;                          // [4][PSLOT]
  int* flags = reinterpret_cast<int*>(Ps + 4 * PSLOT);            // [4 pairs][2]: tile number up to which the partial dP / the dS^T image is ready
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem_b;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int kb = wid & 3, half = wid >> 2;                        // key block of the pair, role (0 = A, 1 = B)
  const int n = blockIdx.y, j0 = ((int)blockIdx.x + kb0) * 128 + kb * 32;   // this pair's keys
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;
  const float* g1b = g1 + (int64_t)n * T * C1;
  const float* g2b = g2 ? g2 + (int64_t)n * T * C2 : nullptr;
  const float* lseb = lse + (int64_t)n * T;
  const float* delb = delta + (int64_t)n * T;

  unsigned char* ks = Ks + kb * 3 * KIMG;
  for (int f = half * 64 + lane; f < 8 * D; f += 128) {  // the pair cuts its key block into pieces together
    const int key = f / (D / 4), dq4 = f % (D / 4);
    const float4 v = *reinterpret_cast<const float4*>(qb + (int64_t)(j0 + key) * D + dq4 * 4);
    uint32_t a0, a1, a2, b0, b1, b2;
    split3_pair(v.x, v.y, a0, a1, a2);
    split3_pair(v.z, v.w, b0, b1, b2);
    unsigned char* d = ks + key * KP + dq4 * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(d + KIMG) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + 2 * KIMG) = make_uint2(a2, b2);
  }
  // B fragments of this wave's keys over ITS channel half: V[key = l31][half * CT/2 + 16 kk + 8 lh + j]
  float vfrag[CT / 32][8];
#pragma unroll
  for (int kk = 0; kk < CT / 32; ++kk) {
    const int c = half * (CT / 2) + 16 * kk + 8 * lh;
    const float* src = (c < C1) ? v1b + (int64_t)(j0 + l31) * C1 + c : v2b + (int64_t)(j0 + l31) * C2 + (c - C1);
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    vfrag[kk][0] = a.x, vfrag[kk][1] = a.y, vfrag[kk][2] = a.z, vfrag[kk][3] = a.w;
    vfrag[kk][4] = b.x, vfrag[kk][5] = b.y, vfrag[kk][6] = b.z, vfrag[kk][7] = b.w;
  }

  f32x16 acc_dv[CH], acc_dk;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dv[c][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_dk[r] = 0.f;

  float4 rq[NQL], rg[NVL];
  float rl = 0.f, rd = 0.f;
  auto gload = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 512 * i;
      const int row = f / (D / 4), dd = (f % (D / 4)) * 4;
      rq[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(i0 + row) * D + dd) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 512 * i;
      const int row = f / (CT / 4), c = (f % (CT / 4)) * 4;
      rg[i] = (c < C1) ? *reinterpret_cast<const float4*>(g1b + (int64_t)(i0 + row) * C1 + c)
                       : *reinterpret_cast<const float4*>(g2b + (int64_t)(i0 + row) * C2 + (c - C1));
    }
    rl = lseb[i0 + (tid & 31)];
    rd = delb[i0 + (tid & 31)];
  };
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 512 * i;
      if (f < 8 * D) {
        uint32_t a0, a1, a2, b0, b1, b2;
        split3_pair(rq[i].x, rq[i].y, a0, a1, a2);
        split3_pair(rq[i].z, rq[i].w, b0, b1, b2);
        unsigned char* d = Qs + (f / (D / 4)) * QP + (f % (D / 4)) * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(d + QIMG) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(d + 2 * QIMG) = make_uint2(a2, b2);
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 512 * i;
      const int row = f / (CT / 4), c4 = f % (CT / 4), ch = c4 >> 1;
      uint32_t a0, a1, a2, b0, b1, b2;
      split3_pair(rg[i].x, rg[i].y, a0, a1, a2);
      split3_pair(rg[i].z, rg[i].w, b0, b1, b2);
      unsigned char* d = Gs + row * GP + 16 * ((ch & ~15) | ((ch ^ swz(row)) & 15)) + 8 * (c4 & 1);
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + GIMG) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 2 * GIMG) = make_uint2(a2, b2);
    }
    if (tid < 32) {
      lse_i[tid] = rl;
      del_i[tid] = rd;
    }
  };
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lp4;
  typedef __attribute__((address_space(3))) bf16x8_t* lp8;
  typedef __attribute__((address_space(3))) float* lpf;
  auto tr2 = [&](uint32_t a0, uint32_t a1) {
    union {
      s16x4_t h[2];
      bf16x8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a1);
    asm volatile("" : "+v"(u.v));  // LDS return data into an architectural VGPR (see attn_bwd2_x6_kernel)
    return u.v;
  };
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, gb = (lane >> 4) & 1;
  const uint32_t hoff = (uint32_t)(half * (CT / 2) * 2);              // this wave's channel half inside a gO row: + CT bytes
  const uint32_t g_row = lds0 + (uint32_t)(l31 * GP) + hoff;
  const int g_rx = (lh ^ swz(l31)) & 15;
  const uint32_t g_tr = lds0 + (uint32_t)((4 * lh + tq) * GP + 8 * (tp & 1)) + hoff;
  const int g_tx = (2 * gb + (tp >> 1)) ^ ((tq << 2) | lh);
  const uint32_t q_row = lds0 + (uint32_t)(3 * GIMG + l31 * QP + 16 * lh);
  const uint32_t q_tr = lds0 + (uint32_t)(3 * GIMG + (4 * lh + tq) * QP + (16 * gb + 4 * tp) * 2) + (uint32_t)(64 * half);   // d-tile = half
  const uint32_t k_base = lds0 + (uint32_t)(Ks - smem_b) + (uint32_t)(kb * 3 * KIMG);
  const uint32_t k_row = k_base + (uint32_t)(l31 * KP + 16 * lh);
  const uint32_t k_tr = k_base + (uint32_t)((8 * lh + tq) * KP + (16 * gb + 4 * tp) * 2) + (uint32_t)(64 * half);            // d-tile = half
  const uint32_t p_base = lds0 + (uint32_t)(Ps - smem_b) + (uint32_t)(kb * PSLOT);
  const uint32_t t_wr = p_base + (uint32_t)(l31 * 64);
  const int t_wx = (l31 >> 1) & 7;
  const uint32_t t_tr = p_base + (uint32_t)((8 * lh + tq) * 64);
  const int t_rx = (4 * lh) | (tq >> 1);
  float* RBw = reinterpret_cast<float*>(Ps + kb * PSLOT + half * 4096);   // this wave's partial dQ tile [16][64]

  if (tid < 8) flags[tid] = 0;
  // pair-level hand-offs instead of two more workgroup barriers per tile: the producing wave drains its LDS stores and publishes the tile
  // number, the consuming wave polls it (one CU's LDS is coherent and a wave's LDS operations complete in order); the waves of a pair --
  // which share a SIMD -- are then free to drift apart, one running matrix work while the other is in a vector / LDS phase
  const uint32_t f_pdp = lds0 + (uint32_t)((unsigned char*)flags - smem_b) + (uint32_t)(kb * 8), f_img = f_pdp + 4;
  auto publish = [&](uint32_t flag, int it) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *(__attribute__((address_space(3))) volatile int*)(uintptr_t)flag = it;
  };
  auto await = [&](uint32_t flag, int it) {
    while (*(__attribute__((address_space(3))) volatile int*)(uintptr_t)flag < it) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int i0 = 0; i0 < T; i0 += 32) {
    const int it = (i0 >> 5) + 1;
    // the swizzle constants are made opaque once per tile: the ~40 LDS addresses derived from them are then recomputed where they are used
    // (two vector instructions each) instead of being hoisted out of the loop, where they cost 34 spilled registers at the 256-register cap
    int grx = g_rx, gtx = g_tx, twx = t_wx, trx = t_rx;
    asm volatile("" : "+v"(grx), "+v"(gtx), "+v"(twx), "+v"(trx));
    ATT_STAMP(0);
    // ---- S[q][key] (both), partial dP[q][key] over this wave's channel half
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sp[r] = 0.f;
      dp[r] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
      bf16x8_t a[3], b[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) a[pc] = *(lp8)(uintptr_t)(q_row + (uint32_t)(pc * QIMG + kk * 32));
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) b[pc] = *(lp8)(uintptr_t)(k_row + (uint32_t)(pc * KIMG + kk * 32));
      sp = mfma_x6(a, b, sp);
    }
    ATT_STAMP(1);
    {
      auto g_frag = [&](int kk, bf16x8_t (&a)[3]) {
        const uint32_t ad = g_row + (uint32_t)(16 * (((2 * kk) & 15) ^ grx) + 16 * ((2 * kk) & ~15));
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a[pc] = *(lp8)(uintptr_t)(ad + (uint32_t)(pc * GIMG));
      };
      bf16x8_t a[3], b[3];
#pragma unroll
      for (int kk = 0; kk < CT / 32; ++kk) {  // no register prefetch of the next fragment: the partner wave on this SIMD covers the LDS latency
        g_frag(kk, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(vfrag[kk][j]));  // keep the split inside the tile loop (see attn_bwd2_x6_kernel)
        split3_bf16(vfrag[kk], b);
        dp = mfma_x6(a, b, dp);
      }
    }
    ATT_STAMP(2);
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[r] = __expf(sp[r] - lse_i[(r & 3) + 8 * (r >> 2) + 4 * lh]);   // P
    if (half == 0) {  // A (the older wave of the pair: it gets here first): its partial dP tile to the pair slot, then straight on to dV
#pragma unroll
      for (int r = 0; r < 16; ++r) *(lpf)(uintptr_t)(p_base + (uint32_t)((r * 64 + lane) * 4)) = dp[r];
      publish(f_pdp, it);
    }
    ATT_STAMP(3);
    if (half == 1) {  // B: full dP, dS, the transposed dS piece image
      await(f_pdp, it);
      ATT_STAMP(4);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] += *(lpf)(uintptr_t)(p_base + (uint32_t)((r * 64 + lane) * 4));
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = sp[r] * (dp[r] - del_i[(r & 3) + 8 * (r >> 2) + 4 * lh]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the partial tile has returned before the image overwrites it
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t ds[3];
        const float e[8] = {dp[8 * s], dp[8 * s + 1], dp[8 * s + 2], dp[8 * s + 3], dp[8 * s + 4], dp[8 * s + 5], dp[8 * s + 6], dp[8 * s + 7]};
        split3_bf16(e, ds);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(3))) u32x2_t* lpu2;
          const u32x4_t w = __builtin_bit_cast(u32x4_t, ds[pc]);
          *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s) | lh) ^ twx))) = u32x2_t{w[0], w[1]};
          *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s + 2) | lh) ^ twx))) = u32x2_t{w[2], w[3]};
        }
      }
      publish(f_img, it);
    }
    ATT_STAMP(5);
    // ---- dV^T[c][key] += gO^T[c][q] P[q][key] over this wave's channel tiles (B runs this while A is still in the block above)
    auto dv_step = [&](int s) {
      bf16x8_t pp[3];
      const float f[8] = {sp[8 * s], sp[8 * s + 1], sp[8 * s + 2], sp[8 * s + 3], sp[8 * s + 4], sp[8 * s + 5], sp[8 * s + 6], sp[8 * s + 7]};
      split3_bf16(f, pp);
      auto gt_frag = [&](int c, bf16x8_t (&a)[3]) {
        const uint32_t ch0 = (uint32_t)(16 * (((4 * c) & 15) ^ gtx) + 16 * ((4 * c) & ~15));
        const uint32_t ch1 = (uint32_t)(16 * (((4 * c) & 15) ^ gtx ^ 2) + 16 * ((4 * c) & ~15));
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          a[pc] = tr2(g_tr + (uint32_t)(pc * GIMG + 16 * s * GP) + ch0, g_tr + (uint32_t)(pc * GIMG + (16 * s + 8) * GP) + ch1);
      };
      bf16x8_t a[3];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        gt_frag(c, a);
        acc_dv[c] = mfma_x6(a, pp, acc_dv[c]);
      }
    };
    dv_step(0);
    dv_step(1);
    ATT_STAMP(6);
    if (half == 0) await(f_img, it);  // (B wrote the image itself)
    ATT_STAMP(7);
    // the next query tile's global loads start here: the dK / dQ phase below (24 MFMAs per wave) covers their latency; issued right in
    // front of barrier (3) they were waited for in full by the staging stores behind it
    gload(i0 + 32 < T ? i0 + 32 : i0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- dK^T[d][key] += Q^T[d][q] dS[q][key] and dQ[q][d] = dS[q][key] K[key][d], this wave's d-tile; dS from the pair's image
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t dsr[3], a[3], b[3];
      {  // B operand of dK^T: lane (key l31, half lh) needs queries 16 s + 4 lh + 0..3 and 16 s + 8 + 4 lh + 0..3 -- the k order of the
         // transposed Q fragments -- i.e. exactly the two 8-byte slots wave A's lane (l31, lh) stored for this step
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) u32x2_t* lpu2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const u32x2_t lo = *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s) | lh) ^ twx)));
          const u32x2_t hi = *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s + 2) | lh) ^ twx)));
          dsr[pc] = __builtin_bit_cast(bf16x8_t, u32x4_t{lo[0], lo[1], hi[0], hi[1]});
        }
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) a[pc] = tr2(q_tr + (uint32_t)(pc * QIMG + 16 * s * QP), q_tr + (uint32_t)(pc * QIMG + (16 * s + 8) * QP));
      acc_dk = mfma_x6(a, dsr, acc_dk);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        a[pc] = tr2(t_tr + (uint32_t)(pc * TIMG + 16 * s * 64 + 8 * ((4 * gb + tp) ^ trx)),
                    t_tr + (uint32_t)(pc * TIMG + (16 * s + 4) * 64 + 8 * ((4 * gb + tp) ^ (trx + 2))));
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) b[pc] = tr2(k_tr + (uint32_t)(pc * KIMG + 16 * s * KP), k_tr + (uint32_t)(pc * KIMG + (16 * s + 4) * KP));
      dq = mfma_x6(a, b, dq);
    }
    ATT_STAMP(8);
    __syncthreads();  // (3) every read of the dS^T image, of the gO / Q images and of lse / delta is done
    ATT_STAMP(9);
#pragma unroll
    for (int r = 0; r < 16; ++r) RBw[r * 64 + lane] = dq[r];
    lstore();
    ATT_STAMP(10);
    __syncthreads();  // (4) the partial dQ tiles of all pairs and the next query tile are in place
    ATT_STAMP(11);
    // sum over the four key blocks: 2 d-tiles x 16 register rows = 32 rows, wave w takes rows 4 w .. 4 w + 3
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = wid * 4 + rr;  // = c * 16 + r
      const int c = row >> 4, r = row & 15;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += reinterpret_cast<const float*>(Ps + w * PSLOT + c * 4096)[r * 64 + lane];
      atomicAdd(gq + ((int64_t)n * T + i0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + c * 32 + l31, sum);
    }
    ATT_STAMP(12);
    __syncthreads();  // (5) the slots are free for the next tile's partial dP
    ATT_STAMP(13);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: dV rows of this wave's keys and channel half (plain stores), the key-side dQ of its d-tile (atomics)
  const int64_t row = (int64_t)n * T + j0 + l31;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = half * (CT / 2) + c * 32;
    float* ob = (ch < C1) ? gv1 + row * C1 + ch : gv2 + row * C2 + (ch - C1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) = make_float4(acc_dv[c][4 * g], acc_dv[c][4 * g + 1], acc_dv[c][4 * g + 2], acc_dv[c][4 * g + 3]);
  }
  {
    float* gqb = gq + row * D + half * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(gqb + (r & 3) + 8 * (r >> 2) + 4 * lh, acc_dk[r]);
  }
}

