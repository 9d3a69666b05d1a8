// bf16 weight gradient on the eight-phase schedule of conv_bf16_8ph.h (read that file's header for the schedule, the half-image ring,
// the counted waits and the two staggered wave groups -- all of it is the same here):
//   dwf[tap][c][k] += sum over pixels  x[pixel + tap][c] * dy[pixel][k]      (stylegan2/model.py:241-279 under autograd)
// as a GEMM with M = (tap, c) rows, N = k columns and the reduction over pixels -- the SLOW index of both NHWC operands, so
//  * the LDS image of a 32-row group is [64 pixels][32 rows] bf16 (4 KB; a wave instruction of the copy fills 16 pixels x 64 bytes,
//    every copy of a thread belongs to ONE pixel: one pixel decode per reduction tile and thread), and MFMA operands are gathered by
//    ds_read_b64_tr_b16, two per operand (the layout and fragment code of wgrad_bf16_kernel in conv_bf16.hip);
//  * v_mfma_f32_32x32x16_bf16: a quadrant (64 rows x 32 columns) is 2 tiles x 4 reduction steps = 8 MFMAs of 32 cycles -- the
//    same 256 cycles per phase as the forward kernel's 16 x 16 cycles;
//  * half-images by reading phase: A0 = the first two 32-row groups of every wave row (phase 0), B0 = the first column group of every
//    wave column (phase 0), B1 (phase 1), A1 (phase 2): 16 KB each, two LDS-DMA instructions per thread, eight in flight;
//  * the pixel range is split over workgroups (fp32 atomics meet in dwf); consecutive logical workgroups share a split, and the XCD
//    remap keeps them on one L2.
#pragma once

struct Wg8Args {
  const bf16_t* x;
  const bf16_t* dy;
  float* dwf;
  ConvGeom g;  // forward geometry: anchors = output pixels, all kh*kw taps
  int Kout, ycs, Mrows, P, kchunk;
  int tiles, tiles_n, ksplit;
};

__global__ void __launch_bounds__(512) wgrad_bf16_8ph_kernel(Wg8Args a) {
  constexpr int NA = 2, NB = 2, AH = 16384, BH = 16384, STAGE = 65536, WFULL = 2 * NA + 2 * NB;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, grp = wid >> 2;  // 2 x 4 waves, 128 rows x 64 columns each
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / a.tiles, tl = lid - split * a.tiles;
  const int tile_m = tl / a.tiles_n, tile_n = tl - tile_m * a.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const ConvGeom& g = a.g;
  const int k_begin = split * a.kchunk;
  int k_end = k_begin + a.kchunk;
  if (k_end > a.P) k_end = a.P;
  const int nt = (k_end - k_begin + 63) >> 6;
  if (nt <= 0) return;  // whole workgroup

  // ---- copies: wave w fills pixel block w & 3 (16 pixels) of the row groups {w >> 2, (w >> 2) + 2} of every half-image; lane =
  // 4 * pixel + chunk (8 rows of the 32-row group)
  const int pb = wid & 3, gsel = wid >> 2, cq = lane & 3, kr = 16 * pb + (lane >> 2);
  const void* zp = fmi_chunk_zero;
  asm volatile("" : "+s"(zp));
  int a_dy[2][2], a_dx[2][2], a_off[2][2];  // [sub][j]: tap shift (wave-uniform) and element offset of the thread's 8 rows
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ga = gsel + 2 * j;                                  // group inside the half-image: wave row ga >> 1, its group ga & 1
      const int row = m0 + 32 * ((ga >> 1) * 4 + sub * 2 + (ga & 1));
      const int t = (int)fdiv((uint32_t)row, g.dC);
      const int i = (int)fdiv((uint32_t)t, g.dntx), jx = t - i * g.ntx;
      a_dy[sub][j] = row < a.Mrows ? g.dy0 + i : -0x20000000;  // rows past the last tap: never inside the image
      a_dx[sub][j] = g.dx0 + jx;
      a_off[sub][j] = ((g.dy0 + i) * g.IW + a_dx[sub][j]) * g.cstride + (row - t * g.C) + 8 * cq;
    }
  int b_col[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 32 * (2 * (gsel + 2 * j) + sub) + 8 * cq;  // wave column gsel + 2 j, its column group `sub`
      b_col[sub][j] = col < a.Kout ? col : -1;
    }
  // the pixel of this thread in a reduction tile: anchor of its input window and validity
  struct PixAt {
    int iy0, ix0;
    int64_t xb;    // element offset of x[n][iy0][ix0][0]
    int64_t yb;    // element offset of dy[pixel][0]
    bool ok;
  };
  auto pix_at = [&](int k0) {
    PixAt p;
    const int pix = k0 + kr;
    p.ok = pix < k_end;
    const uint32_t n = fdiv((uint32_t)pix, g.dG);
    const uint32_t rem = (uint32_t)pix - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    p.iy0 = (int)gy * g.S, p.ix0 = (int)gx * g.S;
    p.xb = ((int64_t)((int)n * g.IH + p.iy0) * g.IW + p.ix0) * g.cstride;
    p.yb = (int64_t)pix * a.ycs;
    return p;
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma2 = [&](const void* g0, const void* g1, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "s"(dst), "s"(dst + 8192)
                 : "memory");
  };
  // group ga of a half-image lies at ga * 4096, pixel block pb at pb * 1024: this wave's two copies are 8192 bytes apart
  const uint32_t wdst = __builtin_amdgcn_readfirstlane((uint32_t)gsel * 4096u + (uint32_t)pb * 1024u);
  auto issueA = [&](int sub, const PixAt& p, int st) {
    const void* gp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = p.ok && (unsigned)(p.iy0 + a_dy[sub][j]) < (unsigned)g.IH && (unsigned)(p.ix0 + a_dx[sub][j]) < (unsigned)g.IW;
      gp[j] = ok ? (const void*)(a.x + p.xb + a_off[sub][j]) : zp;
    }
    dma2(gp[0], gp[1], lds0 + (uint32_t)(st * STAGE + sub * AH) + wdst);
  };
  auto issueB = [&](int sub, const PixAt& p, int st) {
    const void* gp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) gp[j] = (p.ok && b_col[sub][j] >= 0) ? (const void*)(a.dy + p.yb + b_col[sub][j]) : zp;
    dma2(gp[0], gp[1], lds0 + (uint32_t)(st * STAGE + 2 * AH + sub * BH) + wdst);
  };

  // ---- transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, rows 4p..4p+3 of the group's 16
  const int i16 = lane & 15;
  const uint32_t lbase = (uint32_t)((8 * lh + (i16 >> 2)) * 64 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3));
  const uint32_t a_rd = lds0 + (uint32_t)(wr * 2) * 4096u + lbase;           // + stage + sub * AH + gi * 4096 + s * 1024 (+ 256)
  const uint32_t b_rd = lds0 + 2 * AH + (uint32_t)wc * 4096u + lbase;        // + stage + sub * BH + s * 1024 (+ 256)
  auto frag = [&](uint32_t ad) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)ad);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ad + 256u));
    union {
      s16x4 h[2];
      bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
  };
  bf16x8 fa[2][4], fb0[4], fb1[4];
  auto read_a = [&](int st, int sub) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[gi][s] = frag(a_rd + (uint32_t)(st * STAGE + sub * AH + gi * 4096 + s * 1024));
  };
  auto read_b = [&](int st, int sub, bf16x8 (&fb)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[s] = frag(b_rd + (uint32_t)(st * STAGE + sub * BH + s * 1024));
  };
  f32x16 acc[2][2][2];  // [row half i][column half j][row group gi]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][gi][r] = 0.f;
  auto mfmas = [&](f32x16 (&d)[2], const bf16x8 (&fb)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) d[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gi][s], fb[s], d[gi], 0, 0, 0);
  };
#define FMI_W8_MID()                                  \
  __builtin_amdgcn_s_barrier();                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0);                  \
  __builtin_amdgcn_s_setprio(1)
#define FMI_W8_END()                 \
  __builtin_amdgcn_s_setprio(0);     \
  __builtin_amdgcn_sched_barrier(0); \
  __builtin_amdgcn_s_barrier();      \
  asm volatile("" ::: "memory")
#define FMI_W8_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

  // prologue: tile 0 whole, A0 / B0 of tile 1 -- the order the loop issues in
  PixAt p1 = pix_at(k_begin), p2;
  issueA(0, p1, 0);
  issueB(0, p1, 0);
  issueB(1, p1, 0);
  issueA(1, p1, 0);
  p1 = pix_at(k_begin + 64);
  if (nt > 1) {
    issueA(0, p1, 1);
    issueB(0, p1, 1);
    FMI_W8_WAIT(WFULL);
  } else {
    FMI_W8_WAIT(NA + NB);
  }
  p2 = pix_at(k_begin + 128);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp) __builtin_amdgcn_s_barrier();  // the second four waves run one barrier behind
  int st = 0;
  for (int u = 0; u < nt; ++u) {
    const bool n1 = u + 1 < nt, n2 = u + 2 < nt;
    // phase 0: quadrant (i0, j0)
    read_b(st, 0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(st, 0);
    if (n1) {
      issueB(1, p1, st ^ 1);
      FMI_W8_WAIT(WFULL);
    } else {
      FMI_W8_WAIT(NA);
    }
    FMI_W8_MID();
    mfmas(acc[0][0], fb0);
    FMI_W8_END();
    // phase 1: (i0, j1)
    read_b(st, 1, fb1);
    if (n1) {
      issueA(1, p1, st ^ 1);
      FMI_W8_WAIT(WFULL);
    } else {
      FMI_W8_WAIT(0);
    }
    FMI_W8_MID();
    mfmas(acc[0][1], fb1);
    FMI_W8_END();
    // phase 2: (i1, j1)
    read_a(st, 1);
    if (n2) issueA(0, p2, st);
    FMI_W8_MID();
    mfmas(acc[1][1], fb1);
    FMI_W8_END();
    // phase 3: (i1, j0)
    p1 = p2;
    if (n2) {
      issueB(0, p2, st);
      p2 = pix_at(k_begin + (u + 3) * 64);
      FMI_W8_WAIT(WFULL);
    } else if (n1) {
      FMI_W8_WAIT(NA + NB);
    } else {
      FMI_W8_WAIT(0);
    }
    FMI_W8_MID();
    mfmas(acc[1][0], fb0);
    FMI_W8_END();
    st ^= 1;
  }
  if (!grp) __builtin_amdgcn_s_barrier();
#undef FMI_W8_MID
#undef FMI_W8_END
#undef FMI_W8_WAIT

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 128 + i * 64 + gi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= a.Mrows) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wc * 64 + j * 32 + l31;
          if (col < a.Kout) atomicAdd(a.dwf + (int64_t)row * a.Kout + col, acc[i][j][gi][r]);
        }
      }
}
