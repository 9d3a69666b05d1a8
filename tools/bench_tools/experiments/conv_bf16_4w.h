// SHELVED EXPERIMENT (not built): four-wave, two-workgroups-per-CU sibling of the eight-phase bf16 convolution kernel.  Correct (it passed
// tests/test_gpu_bf16.py::test_conv_bf16_eight_phase_kernel under a forcing mode), NOT faster: forward / adjoint TFLOP/s on one box,
// 512->512 at 64^2 | 256->256 at 128^2 | 128->128 at 256^2 (batch 16): eight-phase kernels 1018 1148 | 1106 1127 | 861 858; this kernel
// 932 1022 | 1013 1028 | 861 867; the decoder's 39 convolution launches 5.70 ms with the eight-phase kernels only, 5.78 with this one on the
// 128-channel layers and the short-reduction phases.  So two co-resident workgroups do not buy back the work per output tile: each wave
// issues 12 DMA instructions per reduction tile instead of 8, the weight half-images are single-buffered, and a workgroup's own four waves
// (one per SIMD) no longer alternate reads and MFMAs against a partner.
// Four-wave sibling of conv_bf16_8ph.h for what that kernel leaves exposed: the work per OUTPUT tile (DMA prologue, output stage, store
// burst: 13-24 % of a launch at 18-36 reduction tiles per output tile, more of the ConvTranspose phases with 4-16).  One workgroup of
// eight waves per CU has nothing to overlap it with; here a workgroup is FOUR waves (2 x 2, each 128 pixels x 64 channels, tile 256 x 128)
// on 80 KB of LDS, so TWO workgroups share a CU: one's prologue / output stage runs under the other's MFMAs, and the two waves of a SIMD
// belong to different workgroups that drift apart by themselves (no stagger barrier).  Same operand images, swizzle, quadrant order and
// per-row contexts as the eight-wave kernel; what differs:
//  * LDS: the pixel operand's half-images A0 / A1 are double-buffered (2 x 32 KB), the weight operand's B0 / B1 single (16 KB): B0 is
//    read in phase 0 and refilled in phase 1, B1 read in phase 1 and refilled in phase 2 (weights come from L2);
//  * DMA stream per reduction tile u: phase 0 A1(u+1), phase 1 B0(u+1), phase 2 B1(u+1), phase 3 A0(u+2); counted waits at the END of a
//    phase for what the next one reads: vmcnt(8) / (14) / - / (6);
//  * one barrier per phase in front of the reads (the wait sits before it, the refill of a half-image is issued behind the barrier that
//    follows its reading phase); phase 3 reads nothing and needs none: three barriers per reduction tile.
#pragma once

__global__ void __launch_bounds__(256, 2) conv_bf16_4w_kernel(ConvSetB set, EpActB act) {
  constexpr int BM = 256, BN = 128, NA = 4, NB = 2, AH = 16384, BH = 8192, ASTAGE = 2 * AH;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  if (lid >= set.ph[blockIdx.y].tiles) return;  // whole workgroup
  const ConvKB la = set.ph[blockIdx.y].la;
  const ConvWKB lb = set.ph[blockIdx.y].lb;
  const int K = set.ph[blockIdx.y].K, tiles_n = set.tiles_n;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * ASTAGE + 2 * BH];  // A0 A1 | A0 A1 | B0 B1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging contexts (as in the eight-wave kernel): thread tid fills chunk position tid & 7 of local rows (tid >> 3) + 32 j
  const ConvGeom& g = la.g;
  const int r0 = tid >> 3, kq = (((tid & 7) ^ ((tid >> 4) & 7)) << 3);
  const void* zp = fmi_chunk_zero;
  asm volatile("" : "+s"(zp));
  const unsigned char* pa[2][NA];
  uint32_t ma[2][NA];
  const unsigned char* pb[2][NB];
  uint32_t mb[2][NB];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int lr = r0 + 32 * j;  // wave row lr >> 6, row lr & 63 of its sub-half
      const ConvKB::DCtx d = la.dprep(m0 + (lr >> 6) * 128 + sub * 64 + (lr & 63), kq);
      pa[sub][j] = reinterpret_cast<const unsigned char*>(la.p + d.boff);
      uint32_t ym = 0, xm = 0;  // taps whose row / column lies inside the image
      for (int i = 0; i < g.nty; ++i)
        if ((unsigned)(d.ry + g.ystep * i) < (unsigned)g.IH) ym |= 1u << i;
      for (int jj = 0; jj < g.ntx; ++jj)
        if ((unsigned)(d.rx + g.xstep * jj) < (unsigned)g.IW) xm |= 1u << jj;
      uint32_t m = 0;
      for (int i = 0; i < g.nty; ++i)
        if (ym >> i & 1) m |= xm << (i * g.ntx);
      ma[sub][j] = m;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int lr = r0 + 32 * j;  // wave column lr >> 5, column lr & 31 of its sub-half
      const ConvWKB::DCtx d = lb.dprep(n0 + (lr >> 5) * 64 + sub * 32 + (lr & 31), kq);
      pb[sub][j] = d.off >= 0 ? reinterpret_cast<const unsigned char*>(lb.p + d.off) : reinterpret_cast<const unsigned char*>(zp);
      mb[sub][j] = d.off >= 0 ? 0xffffffffu : 0u;
    }
  }

  const int ntx = g.ntx, nty = g.nty;
  const int a_dx = g.xstep * g.cstride * 2, a_dy = (g.ystep * g.IW - (ntx - 1) * g.xstep) * g.cstride * 2;
  const int a_dc = 128 - ((nty - 1) * g.ystep * g.IW + (ntx - 1) * g.xstep) * g.cstride * 2;
  const int b_dx = g.kwstep * g.C * 2, b_dy = (g.khstep * g.kw - (ntx - 1) * g.kwstep) * g.C * 2;
  const int b_dc = 128 - ((nty - 1) * g.khstep * g.kw + (ntx - 1) * g.kwstep) * g.C * 2;
  struct TileAt {
    int ti, tj;
    uint32_t bit;
    int64_t ua;
    uint32_t ub;
  };
  auto advance = [&](TileAt& t) {
    if (++t.tj == ntx) {
      t.tj = 0;
      if (++t.ti == nty) t.ti = 0, t.bit = 1u, t.ua += a_dc, t.ub += (uint32_t)b_dc;
      else t.bit <<= 1, t.ua += a_dy, t.ub += (uint32_t)b_dy;
    } else {
      t.bit <<= 1, t.ua += a_dx, t.ub += (uint32_t)b_dx;
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma4 = [&](const void* g0, const void* g1, const void* g2, const void* g3, uint32_t dst) {  // four pieces 4 KB apart
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(dst), "s"(dst + 4096), "s"(dst + 8192), "s"(dst + 12288)
                 : "memory");
  };
  auto dma2 = [&](const void* g0, const void* g1, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "s"(dst), "s"(dst + 4096)
                 : "memory");
  };
  auto issueA = [&](int sub, const TileAt& t, int st) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * ASTAGE + sub * AH) + (uint32_t)wid * 1024u);
    const void* gp[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) gp[j] = (ma[sub][j] & t.bit) ? (const void*)(pa[sub][j] + t.ua) : zp;
    dma4(gp[0], gp[1], gp[2], gp[3], dst);
  };
  auto issueB = [&](int sub, const TileAt& t) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(2 * ASTAGE + sub * BH) + (uint32_t)wid * 1024u);
    dma2((const void*)(pb[sub][0] + (t.ub & mb[sub][0])), (const void*)(pb[sub][1] + (t.ub & mb[sub][1])), dst);
  };

  // ---- fragment read addresses
  const int l15 = lane & 15, c0 = (lane >> 4) ^ ((lane >> 1) & 7);
  const uint32_t a_off0 = (uint32_t)(wr * 8192 + l15 * 128 + c0 * 16), a_off1 = a_off0 ^ 64u;
  const uint32_t b_off0 = (uint32_t)(2 * ASTAGE + wc * 4096 + l15 * 128 + c0 * 16), b_off1 = b_off0 ^ 64u;

  f32x4v acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[i][j][r][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 ax[4][2] = {}, bw0[2][2] = {}, bw1[2][2] = {};

  const int nt = K >> 6;
  auto read_a = [&](int st, int sub) {
    const unsigned char* p0 = lds + st * ASTAGE + sub * AH + a_off0;
    const unsigned char* p1 = lds + st * ASTAGE + sub * AH + a_off1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ax[r][0] = *reinterpret_cast<const bf16x8*>(p0 + r * 2048);
      ax[r][1] = *reinterpret_cast<const bf16x8*>(p1 + r * 2048);
    }
  };
  auto read_b = [&](int sub, bf16x8 (&bw)[2][2]) {
    const unsigned char* p0 = lds + sub * BH + b_off0;
    const unsigned char* p1 = lds + sub * BH + b_off1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bw[c][0] = *reinterpret_cast<const bf16x8*>(p0 + c * 2048);
      bw[c][1] = *reinterpret_cast<const bf16x8*>(p1 + c * 2048);
    }
  };
  auto mfmas = [&](f32x4v (&d)[4][2], const bf16x8 (&bw)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) d[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[c][s], ax[r][s], d[r][c], 0, 0, 0);
  };
#define FMI_4W_BAR()            \
  __builtin_amdgcn_s_barrier(); \
  asm volatile("" ::: "memory")
#define FMI_4W_GO()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0);                  \
  __builtin_amdgcn_s_setprio(1)
#define FMI_4W_DONE()            \
  __builtin_amdgcn_s_setprio(0); \
  __builtin_amdgcn_sched_barrier(0)
#define FMI_4W_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

  if (nt > 0) {
    TileAt t1{0, 0, 1u, 0, 0}, t2;
    t1.ub = (uint32_t)((g.kh0 * g.kw + g.kw0) * g.C) * 2u;
    issueA(0, t1, 0);
    issueA(1, t1, 0);
    issueB(0, t1);
    issueB(1, t1);
    advance(t1);  // t1: the tile phases 0-2 fill (u + 1); t2: the tile phase 3 fills (u + 2)
    if (nt > 1) {
      issueA(0, t1, 1);
      FMI_4W_WAIT(NB + NA);
    } else {
      FMI_4W_WAIT(NB);
    }
    t2 = t1;
    advance(t2);
    int st = 0;
    for (int u = 0; u < nt; ++u) {
      const bool n1 = u + 1 < nt, n2 = u + 2 < nt;
      // phase 0: quadrant (i0, j0); fills A1 of tile u + 1
      FMI_4W_BAR();
      read_b(0, bw0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(st, 0);
      if (n1) issueA(1, t1, st ^ 1);
      FMI_4W_GO();
      mfmas(acc[0][0], bw0);
      FMI_4W_DONE();
      if (n1) {
        FMI_4W_WAIT(2 * NA);
      } else {
        FMI_4W_WAIT(0);
      }
      // phase 1: (i0, j1); refills B0 (read in phase 0)
      FMI_4W_BAR();
      read_b(1, bw1);
      if (n1) issueB(0, t1);
      FMI_4W_GO();
      mfmas(acc[0][1], bw1);
      FMI_4W_DONE();
      if (n1) {
        FMI_4W_WAIT(3 * NB + 2 * NA);  // B0(u) B1(u) A0(u+1) A1(u+1) B0(u+1) were issued behind A1(u): 2 + 2 + 4 + 4 + 2
      } else {
        FMI_4W_WAIT(2 * NB);
      }
      // phase 2: (i1, j1); refills B1 (read in phase 1)
      FMI_4W_BAR();
      read_a(st, 1);
      if (n1) issueB(1, t1);
      FMI_4W_GO();
      mfmas(acc[1][1], bw1);
      FMI_4W_DONE();
      // phase 3: (i1, j0); no LDS read, no barrier; fills A0 of tile u + 2 (its stage's A0 was read in phase 0)
      if (n2) issueA(0, t2, st);
      t1 = t2;
      advance(t2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mfmas(acc[1][0], bw0);
      FMI_4W_DONE();
      if (n1) {
        if (n2) {
          FMI_4W_WAIT(NB + NA);
        } else {
          FMI_4W_WAIT(NB);
        }
      } else {
        FMI_4W_WAIT(0);
      }
      st ^= 1;
    }
  }
#undef FMI_4W_BAR
#undef FMI_4W_GO
#undef FMI_4W_DONE
#undef FMI_4W_WAIT

  // ---- output: lane = pixel (l & 15) of each 16-pixel group, four consecutive channels 4 (l >> 4) .. + 3 of each 16-channel group
  const ConvEpB ep = set.ph[blockIdx.y].ep;
  const int M = set.ph[blockIdx.y].M, N = set.N;
  const int cl = 4 * (lane >> 4);
  if (false) {  // timing: no output stage at all (one store that keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) t += acc[i][j][r][c][0] + acc[i][j][r][c][1] + acc[i][j][r][c][2] + acc[i][j][r][c][3];
    if (t == 123.25f) ep.y[0] = 1;
    return;
  }
  const float nwv = (act.on && act.noise) ? act.nw[0] : 0.f;
  // a plain stride-1 convolution writes anchor row r to pixel r: no decode of (sample, y, x) per row -- eight of them per lane were a
  // quarter of this stage's instructions; the sample index is only needed for the per-sample column scale
  const bool linear = ep.OS == 1 && ep.GH == ep.OHt && ep.GW == ep.OWt;
  // the lane's four column groups are the same for all of its eight rows: bias once, the per-sample column scale once per sample
  // (a tile straddles a sample boundary at most once) -- reloaded per (row, group) they were 64 loads per lane
  float4 bsv[2][2], csv[2][2];
  int n_cached = -1;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = n0 + wc * 64 + j * 32 + c * 16 + cl;
      bsv[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
      csv[j][c] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (act.on && act.bias && col < N) bsv[j][c] = *reinterpret_cast<const float4*>(act.bias + col);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 128 + i * 64 + r * 16 + l15;
      if (row >= M) continue;
      int n_s = 0;
      int64_t pix;
      if (linear) {
        pix = row;
        if (ep.colscale) n_s = (int)fdiv((uint32_t)row, ep.dG);
      } else {
        pix = ep.row_pix(row, n_s);
      }
      const int64_t off = pix * ep.cstride;
      if (ep.colscale && n_s != n_cached) {
        n_cached = n_s;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = n0 + wc * 64 + j * 32 + c * 16 + cl;
            if (col < N) csv[j][c] = *reinterpret_cast<const float4*>(ep.colscale + (int64_t)n_s * ep.Nout + col);
          }
      }
      float nz = 0.f;
      if (act.on && act.noise) nz = nwv * act.noise[pix];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int col = n0 + wc * 64 + j * 32 + c * 16 + cl;
          if (col >= N) continue;
          f32x4v a = acc[i][j][r][c];
          if (ep.colscale) a[0] *= csv[j][c].x, a[1] *= csv[j][c].y, a[2] *= csv[j][c].z, a[3] *= csv[j][c].w;
          if (act.on) {
            const float bb[4] = {bsv[j][c].x, bsv[j][c].y, bsv[j][c].z, bsv[j][c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = a[e] + nz + bb[e];
              a[e] = (v < 0.f ? v * act.slope : v) * act.gain;
            }
          }
          const bf16x2v lo = __builtin_convertvector((f32x2v){a[0], a[1]}, bf16x2v), hi = __builtin_convertvector((f32x2v){a[2], a[3]}, bf16x2v);
          uint2 v;
          v.x = *reinterpret_cast<const uint32_t*>(&lo);
          v.y = *reinterpret_cast<const uint32_t*>(&hi);
          if (true) *reinterpret_cast<uint2*>(ep.y + off + col) = v;
        }
      }
    }
  }
}
