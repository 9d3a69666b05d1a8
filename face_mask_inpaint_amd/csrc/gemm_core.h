// fp32 matrix-core GEMM core for gfx950, shared by the dense batched GEMM, the implicit-GEMM
// convolution family (forward / adjoint / weight-gradient) and the attention products.
//
// Machine mapping (MI355X_MICROARCH.md, cdna_hip_programming.md section 3 "FP32-input MFMA"):
//  * v_mfma_f32_32x32x2_f32: one wave owns 32x32 output tiles, 16 accumulator registers each;
//    A operand lane l = A[i = l&31][k = l>>5], B operand lane l = B[k = l>>5][j = l&31];
//    D register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].  Exact f32 (fmaf chain).
//  * 256-thread workgroups = 4 waves (one per SIMD), 2 workgroups per CU; operand tiles are staged
//    k-major in LDS ([k][m] with a +4 float row pad) so every fragment read is a conflict-free
//    ds_read_b32 of 32 consecutive floats per half-wave; global->register->LDS double buffering
//    with ONE barrier per 16-deep k-step (register prefetch of tile t+1 is issued before the MFMAs
//    of tile t and written to the other LDS buffer after them).
//  * workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
#pragma once
#include "common.h"
#include <cstdio>
#include <cstdlib>
extern "C" long fmi_debug_launch_counter;
#include <type_traits>

#ifndef FMI_HOST_EMU
#include "x6.h"
#endif

// ---- division by a runtime-invariant 32-bit divisor (Granlund-Montgomery round-up method) ----
struct FastDiv {
  uint32_t m, s1, s2, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__host__ __device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = __umulhi(f.m, n);
#else
  const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

#ifndef FMI_EXP
#define FMI_EXP 0  // bit mask of timing experiments (results become wrong): 1 no barrier, 2 no global loads, 4 no LDS stores in the main loop, 8 loads re-read the first tile; 16 = prefetch distance 2 (results stay right)
#endif

// ---- tile configurations ----
template <int WM_, int WN_, int TM_, int TN_>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
  static constexpr int LDA = BM + 4, LDB = BN + 4;
  static_assert(WM * WN == 4, "4 waves per workgroup");
};
using Tile128x128 = TileCfg<2, 2, 2, 2>;
using Tile128x64 = TileCfg<2, 2, 2, 1>;
using Tile128x32 = TileCfg<4, 1, 1, 1>;
using Tile64x64 = TileCfg<2, 2, 1, 1>;
using Tile64x128 = TileCfg<2, 2, 1, 2>;
using Tile32x128 = TileCfg<1, 4, 1, 1>;

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// =====================================================================================
// Operand loaders.  KMODE: load4(slot, x, k) returns elements (x, k..k+3); otherwise
// (x..x+3, k).  "x" is the row (A) / column (B) index, "k" the reduction index.  Out-of-range
// elements read as zero.  prep(slot, x) lets a loader cache per-row state (x is fixed per slot).
// =====================================================================================
struct DenseK {  // reduction index contiguous in memory
  static constexpr bool KMODE = true;
  const float* p;
  int64_t ld, bs;
  int X, K, vec;
  __device__ __forceinline__ void set_batch(int b) { p += (int64_t)b * bs; }
  struct Ctx {};
  __device__ __forceinline__ Ctx prep(int) const { return Ctx{}; }
  __device__ __forceinline__ float4 load4(const Ctx&, int x, int k) const {
    if (x >= X || k >= K) return zero4();
    const float* q = p + (int64_t)x * ld + k;
    if (vec && k + 3 < K) return ldg4(q);
    float4 r = zero4();
    r.x = q[0];
    if (k + 1 < K) r.y = q[1];
    if (k + 2 < K) r.z = q[2];
    if (k + 3 < K) r.w = q[3];
    return r;
  }
  // LDS-DMA path: address of the 16-byte chunk (x, k..k+3), nullptr = zeros
  // LDS-DMA path.  dprep(x, kq): per-thread constants of one copy slot (row x, k offset kq inside the tile);
  // tile(k0): wave-uniform state of a 16-deep tile; chunk(): address of the 16-byte chunk, nullptr = zeros.
  __host__ __device__ __forceinline__ bool dma_ok() const { return vec && (K & 3) == 0; }
  struct DCtx {
    int64_t off;
    int kq;
  };
  struct Tile {
    const float* base;
    int krem;
  };
  __device__ __forceinline__ DCtx dprep(int x, int kq) const { return DCtx{(int64_t)x * ld + kq, x < X ? kq : 0x40000000}; }
  __device__ __forceinline__ Tile tile(int k0) const { return Tile{p + k0, K - k0}; }
  __device__ __forceinline__ const float* chunk(const DCtx& d, const Tile& t) const { return d.kq < t.krem ? t.base + d.off : nullptr; }
  __device__ __forceinline__ void dstart(DCtx&, int) const {}
  __device__ __forceinline__ void advance(DCtx&) const {}
};

struct DenseX {  // row/column index contiguous in memory
  static constexpr bool KMODE = false;
  const float* p;
  int64_t ld, bs;
  int X, K, vec;
  __device__ __forceinline__ void set_batch(int b) { p += (int64_t)b * bs; }
  struct Ctx {};
  __device__ __forceinline__ Ctx prep(int) const { return Ctx{}; }
  __device__ __forceinline__ float4 load4(const Ctx&, int x, int k) const {
    if (x >= X || k >= K) return zero4();
    const float* q = p + (int64_t)k * ld + x;
    if (vec && x + 3 < X) return ldg4(q);
    float4 r = zero4();
    r.x = q[0];
    if (x + 1 < X) r.y = q[1];
    if (x + 2 < X) r.z = q[2];
    if (x + 3 < X) r.w = q[3];
    return r;
  }
  __host__ __device__ __forceinline__ bool dma_ok() const { return vec && (X & 3) == 0; }
  struct DCtx {
    int64_t off;
    int kr;
  };
  struct Tile {
    const float* base;
    int krem;
  };
  __device__ __forceinline__ DCtx dprep(int x, int kr) const { return DCtx{(int64_t)kr * ld + x, x < X ? kr : 0x40000000}; }
  __device__ __forceinline__ Tile tile(int k0) const { return Tile{p + (int64_t)k0 * ld, K - k0}; }
  __device__ __forceinline__ const float* chunk(const DCtx& d, const Tile& t) const { return d.kr < t.krem ? t.base + d.off : nullptr; }
  __device__ __forceinline__ void dstart(DCtx&, int) const {}
  __device__ __forceinline__ void advance(DCtx&) const {}
};

// Geometry of one implicit-GEMM launch.  Rows enumerate a grid [N][GH][GW] of "anchor" positions;
// tap t = (i, j), i < nty, j < ntx, reads the image at (gy*S + dy0 + ystep*i, gx*S + dx0 + xstep*j)
// and uses weight tap (kh0 + khstep*i)*kw + (kw0 + kwstep*j).
//   forward conv : anchors = output pixels, S = stride, dy0 = -pad, steps = +1, all kh*kw taps
//   adjoint, phase (py,px) of stride s: anchors = x pixels (gy*s+py, gx*s+px), image = dy, S = 1,
//                  kh = kh0 + s*i with (py + pad - kh0) % s == 0, dy0 = (py+pad-kh0)/s, ystep = -1
struct ConvGeom {
  int N, IH, IW, C, cstride;  // gathered image (NHWC), C channels used, pixel pitch cstride
  int GH, GW, S;
  int nty, ntx, dy0, dx0, ystep, xstep, kh0, kw0, khstep, kwstep, kw;
  int pad_mode, vec;
  FastDiv dGW, dG, dC, dntx;
  int64_t img_bs;  // batch (per-sample weight mode) stride of the image
  __host__ __device__ __forceinline__ int ntaps() const { return nty * ntx; }
  __host__ __device__ __forceinline__ int Kdim() const { return nty * ntx * C; }
  __host__ __device__ __forceinline__ int Mdim() const { return N * GH * GW; }
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

struct ConvK {  // A operand of forward / adjoint convolution: gathered pixels x (tap, channel)
  static constexpr bool KMODE = true;
  const float* p;
  ConvGeom g;
  struct Ctx {  // cached anchor of one staged row
    int ry, rx, rn;
  };
  __device__ __forceinline__ void set_batch(int b) { p += (int64_t)b * g.img_bs; }
  __device__ __forceinline__ Ctx prep(int x) const {
    Ctx c{0, 0, -1};
    if (x >= g.Mdim()) return c;
    const uint32_t n = fdiv((uint32_t)x, g.dG);
    const uint32_t rem = (uint32_t)x - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    c.rn = (int)n;
    c.ry = (int)gy * g.S + g.dy0;
    c.rx = (int)gx * g.S + g.dx0;
    return c;
  }
  __device__ __forceinline__ const float* pixel(const Ctx& c, int t, bool& ok) const {
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    int iy = c.ry + g.ystep * i, ix = c.rx + g.xstep * j;
    if (g.pad_mode) {
      iy = reflect_idx(iy, g.IH);
      ix = reflect_idx(ix, g.IW);
    }
    ok = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
    return p + ((int64_t)(c.rn * g.IH + iy) * g.IW + ix) * g.cstride;
  }
  __device__ __forceinline__ float elem(const Ctx& c, int k) const {
    const int t = (int)fdiv((uint32_t)k, g.dC);
    if (t >= g.ntaps()) return 0.f;
    bool ok;
    const float* q = pixel(c, t, ok);
    return ok ? q[k - t * g.C] : 0.f;
  }
  __device__ __forceinline__ float4 load4(const Ctx& c, int, int k) const {
    if (c.rn < 0) return zero4();
    if (g.vec) {  // C % 4 == 0: the four k's share a tap
      const int t = (int)fdiv((uint32_t)k, g.dC);
      if (t >= g.ntaps()) return zero4();
      bool ok;
      const float* q = pixel(c, t, ok);
      return ok ? ldg4(q + (k - t * g.C)) : zero4();
    }
    return make_float4(elem(c, k), elem(c, k + 1), elem(c, k + 2), elem(c, k + 3));
  }
  // LDS-DMA path: only when a 16-deep tile always lies inside one tap (C % 16 == 0) and padding is zeros; the tap is then
  // wave-uniform (scalar decode per tile) and a copy's address is one 64-bit add of a per-thread and a per-tile offset.
  __host__ __device__ __forceinline__ bool dma_ok() const { return g.vec != 0 && (g.C & 15) == 0 && !g.pad_mode; }
  struct DCtx {
    int64_t boff;  // element offset of (anchor pixel, channel kq); the tap adds a wave-uniform offset
    int ry, rx;    // anchor coordinates; rows beyond M get ry far outside the image
  };
  struct Tile {
    int dy, dx;    // tap displacement; past the last tap: dy far outside the image (zeros)
    int64_t uoff;
  };
  __device__ __forceinline__ DCtx dprep(int x, int kq) const {
    const Ctx c = prep(x);
    DCtx d;
    d.ry = c.rn < 0 ? -0x20000000 : c.ry;
    d.rx = c.rx;
    d.boff = ((int64_t)((c.rn < 0 ? 0 : c.rn) * g.IH + c.ry) * g.IW + c.rx) * g.cstride + kq;
    return d;
  }
  __device__ __forceinline__ Tile tile(int k0) const {
    const int tp = (int)fdiv((uint32_t)k0, g.dC);
    const int i = (int)fdiv((uint32_t)tp, g.dntx), j = tp - i * g.ntx;
    Tile t;
    t.dy = tp < g.ntaps() ? g.ystep * i : -0x20000000;
    t.dx = g.xstep * j;
    t.uoff = ((int64_t)(g.ystep * i) * g.IW + t.dx) * g.cstride + (k0 - tp * g.C);
    return t;
  }
  __device__ __forceinline__ const float* chunk(const DCtx& d, const Tile& t) const {
    const bool ok = (unsigned)(d.ry + t.dy) < (unsigned)g.IH && (unsigned)(d.rx + t.dx) < (unsigned)g.IW;
    return ok ? p + d.boff + t.uoff : nullptr;
  }
  __device__ __forceinline__ void dstart(DCtx&, int) const {}
  __device__ __forceinline__ void advance(DCtx&) const {}
};

struct ConvWX {  // B operand of forward / adjoint convolution: packed weights [tap][Cred][Nout]
  static constexpr bool KMODE = false;
  static constexpr bool SPLIT3 = false;
  const float* p;
  ConvGeom g;  // only the tap algebra and C are used
  int64_t bs;
  int Nout, vec;
  struct Ctx {};
  __device__ __forceinline__ void set_batch(int b) { p += (int64_t)b * bs; }
  __device__ __forceinline__ Ctx prep(int) const { return Ctx{}; }
  __device__ __forceinline__ const float* rowp(int k, bool& ok) const {
    const int t = (int)fdiv((uint32_t)k, g.dC);
    ok = t < g.ntaps();
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    return p + ((int64_t)wtap * g.C + (k - t * g.C)) * Nout;
  }
  __device__ __forceinline__ float4 load4(const Ctx&, int x, int k) const {
    bool ok;
    const float* q = rowp(k, ok);
    if (!ok || x >= Nout) return zero4();
    q += x;
    if (vec && x + 3 < Nout) return ldg4(q);
    float4 r = zero4();
    r.x = q[0];
    if (x + 1 < Nout) r.y = q[1];
    if (x + 2 < Nout) r.z = q[2];
    if (x + 3 < Nout) r.w = q[3];
    return r;
  }
  __host__ __device__ __forceinline__ bool dma_ok() const { return vec && (Nout & 3) == 0 && (g.C & 15) == 0; }
  struct DCtx {
    int off;  // kr*Nout + x, or -1 for columns beyond Nout
  };
  struct Tile {
    const float* base;  // row of (tap, first channel of the tile); nullptr past the last tap
  };
  __device__ __forceinline__ DCtx dprep(int x, int kr) const { return DCtx{x < Nout ? kr * Nout + x : -1}; }
  __device__ __forceinline__ Tile tile(int k0) const {
    bool ok;
    const float* q = rowp(k0, ok);
    return Tile{ok ? q : nullptr};
  }
  __device__ __forceinline__ const float* chunk(const DCtx& d, const Tile& t) const { return (d.off >= 0 && t.base) ? t.base + d.off : nullptr; }
  __device__ __forceinline__ void dstart(DCtx&, int) const {}
  __device__ __forceinline__ void advance(DCtx&) const {}
};

#ifndef FMI_HOST_EMU
// B operand of forward / adjoint convolution from weights the preparation kernel already cut into bf16 pieces (x6.h):
//   p3[piece][tap][Cred / 8][Nout][8]   -- one 16-byte chunk = 8 consecutive reduction channels of one output column, which IS the
// lane's B fragment of v_mfma_f32_32x32x16_bf16; a 16-deep tile of a piece is [2 channel groups][BN] chunks, copied lane-linear by the
// LDS-DMA and read back with one conflict-free ds_read_b128 per fragment.  No split arithmetic for this operand in the main loop.
struct ConvWX3 {
  static constexpr bool KMODE = false;
  static constexpr bool SPLIT3 = true;
  const uint16_t* p3;
  ConvGeom g;  // only the tap algebra and C are used
  int64_t pstride;  // elements per piece image (all taps of the kernel)
  int Nout;
  struct Ctx {};
  __device__ __forceinline__ void set_batch(int) {}
  __device__ __forceinline__ Ctx prep(int) const { return Ctx{}; }
  // register-staged path (FMI_DMA_OFF / FMI_DMA_OFF_RANGE / FMI_EXP & 32 debugging, or a loader pair without LDS-DMA): the fp32 weight is
  // the exact sum of its three pieces, added from the smallest up
  __device__ __forceinline__ float elem(int x, int k) const {
    const int t = (int)fdiv((uint32_t)k, g.dC);
    if (t >= g.ntaps() || x >= Nout) return 0.f;
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    const int c = k - t * g.C;
    const uint16_t* q = p3 + (((int64_t)wtap * g.C + c) >> 3) * Nout * 8 + (int64_t)x * 8 + (c & 7);
    const float x0 = __uint_as_float((uint32_t)q[0] << 16), x1 = __uint_as_float((uint32_t)q[pstride] << 16), x2 = __uint_as_float((uint32_t)q[2 * pstride] << 16);
    return (x2 + x1) + x0;
  }
  __device__ __forceinline__ float4 load4(const Ctx&, int x, int k) const { return make_float4(elem(x, k), elem(x + 1, k), elem(x + 2, k), elem(x + 3, k)); }
  __host__ __device__ __forceinline__ bool dma_ok() const { return (g.C & 15) == 0 && ((uintptr_t)p3 & 15) == 0; }
  struct DCtx {
    int64_t off;  // piece * pstride + (kg * Nout + x) * 8, or -1 for columns beyond Nout
  };
  struct Tile {
    const uint16_t* base;  // chunk row of (tap, first channel group of the tile); nullptr past the last tap
  };
  __device__ __forceinline__ DCtx dprep3(int piece, int kg, int x) const { return DCtx{x < Nout ? piece * pstride + ((int64_t)kg * Nout + x) * 8 : -1}; }
  __device__ __forceinline__ Tile tile(int k0) const {
    const int t = (int)fdiv((uint32_t)k0, g.dC);
    if (t >= g.ntaps()) return Tile{nullptr};
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    return Tile{p3 + (((int64_t)wtap * g.C + (k0 - t * g.C)) >> 3) * Nout * 8};
  }
  __device__ __forceinline__ const void* chunk(const DCtx& d, const Tile& t) const { return (d.off >= 0 && t.base) ? (const void*)(t.base + d.off) : nullptr; }
};
#endif

struct WgradAX {  // A operand of the weight gradient: rows = (tap, channel), reduction = anchors
  static constexpr bool KMODE = false;
  const float* p;
  ConvGeom g;
  int ones_row;  // >= 0: one extra row of ones after the (tap, channel) rows, so that the SAME product also yields the
                 // bias gradient sum_pixels dy[p][k] as an extra output row (no separate reduction pass over dy)
  struct Ctx {  // cached (tap, channel) of the first of the four rows a thread stages
    int t0, c0;
  };
  __device__ __forceinline__ void set_batch(int b) { p += (int64_t)b * g.img_bs; }
  __device__ __forceinline__ Ctx prep(int x) const {
    Ctx c;
    c.t0 = (int)fdiv((uint32_t)x, g.dC);
    c.c0 = x - c.t0 * g.C;
    return c;
  }
  __device__ __forceinline__ float elem_tc(int t, int c, int n, int gy, int gx) const {
    if (t >= g.ntaps()) return 0.f;
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    int iy = gy * g.S + g.dy0 + g.ystep * i, ix = gx * g.S + g.dx0 + g.xstep * j;
    if (g.pad_mode) {
      iy = reflect_idx(iy, g.IH);
      ix = reflect_idx(ix, g.IW);
    }
    if ((unsigned)iy >= (unsigned)g.IH || (unsigned)ix >= (unsigned)g.IW) return 0.f;
    return p[((int64_t)(n * g.IH + iy) * g.IW + ix) * g.cstride + c];
  }
  __device__ __forceinline__ float4 load4(const Ctx& cx, int x, int k) const {
    const int t0 = cx.t0, c0 = cx.c0;
    if (k >= g.Mdim()) return zero4();
    if (x == ones_row) return make_float4(1.f, 0.f, 0.f, 0.f);  // ones_row is a multiple of 4 whenever it is enabled
    const uint32_t n = fdiv((uint32_t)k, g.dG);
    const uint32_t rem = (uint32_t)k - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    if (g.vec) {
      if (t0 >= g.ntaps()) return zero4();
      const int i = (int)fdiv((uint32_t)t0, g.dntx), j = t0 - i * g.ntx;
      int iy = (int)gy * g.S + g.dy0 + g.ystep * i, ix = (int)gx * g.S + g.dx0 + g.xstep * j;
      if (g.pad_mode) {
        iy = reflect_idx(iy, g.IH);
        ix = reflect_idx(ix, g.IW);
      }
      if ((unsigned)iy >= (unsigned)g.IH || (unsigned)ix >= (unsigned)g.IW) return zero4();
      return ldg4(p + ((int64_t)((int)n * g.IH + iy) * g.IW + ix) * g.cstride + c0);
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int xe = x + e;
      const int t = (int)fdiv((uint32_t)xe, g.dC);
      v[e] = elem_tc(t, xe - t * g.C, (int)n, (int)gy, (int)gx);
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  __host__ __device__ __forceinline__ bool dma_ok() const { return g.vec != 0; }
  struct DCtx {
    Ctx c;
    int x, kr, dy, dx;
    int k, n, gy, gx;  // anchor (reduction index) this slot fetches in the NEXT tile: walked incrementally, 16 anchors per tile
  };
  struct Tile {
    int k0;
  };
  __device__ __forceinline__ void dstart(DCtx& d, int k_begin) const {
    d.k = k_begin + d.kr;
    const uint32_t n = fdiv((uint32_t)d.k, g.dG);
    const uint32_t rem = (uint32_t)d.k - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    d.n = (int)n;
    d.gy = (int)gy;
    d.gx = (int)(rem - gy * (uint32_t)g.GW);
  }
  __device__ __forceinline__ void advance(DCtx& d) const {
    d.k += 16;
    d.gx += 16;
    while (d.gx >= g.GW) {
      d.gx -= g.GW;
      if (++d.gy == g.GH) {
        d.gy = 0;
        ++d.n;
      }
    }
  }
  __device__ __forceinline__ DCtx dprep(int x, int kr) const {
    DCtx d;
    d.c = prep(x);
    d.x = x;
    d.kr = kr;
    const int i = (int)fdiv((uint32_t)d.c.t0, g.dntx), j = d.c.t0 - i * g.ntx;
    d.dy = g.dy0 + g.ystep * i;
    d.dx = g.dx0 + g.xstep * j;
    return d;
  }
  __device__ __forceinline__ Tile tile(int k0) const { return Tile{k0}; }
  __device__ __forceinline__ const float* chunk(const DCtx& d, const Tile& t) const;  // defined after the constant chunks below
};

// =====================================================================================
// Epilogues: row_off(row) -> element offset of the row, then store(off, col, acc).
// =====================================================================================
struct DenseEp {
  static constexpr bool HAS_VEC4 = false;
  float* c;
  const float* bias;
  int64_t sc_m, sc_n, bs;
  float alpha, beta;
  int atomic;
  __device__ __forceinline__ void set_batch(int b) { c += (int64_t)b * bs; }
  __device__ __forceinline__ int64_t row_off(int row) const { return (int64_t)row * sc_m; }
  __device__ __forceinline__ void store(int64_t off, int col, float v) const {
    float* q = c + off + (int64_t)col * sc_n;
    v *= alpha;
    if (bias) v += bias[col];
    if (atomic) {
      atomicAdd(q, v);
      return;
    }
    if (beta != 0.f) v += beta * *q;
    *q = v;
  }
};

struct ConvEp {  // rows = anchors of the launch geometry, written at (gy*OS+py, gx*OS+px) of [N][OHt][OWt]
  float* y;
  const float* bias;
  const float* res;
  int GH, GW, OS, py, px, OHt, OWt, cstride, act;
  FastDiv dGW, dG;
  int64_t bs;
  int vec = 0;  // 16-byte epilogue allowed: N % 4 == 0, cstride % 4 == 0, y / res / bias 16-byte aligned, not the atomic mode
  // adjoint of act(x) -> conv: the result is multiplied by act'(x) = (mask > 0 ? 1 : mslope), mask = x or act(x) in y's layout
  // (LeakyReLU -> conv of the ResBlocks, ReLU -> conv inside VGG16): the separate activation-backward pass disappears
  const float* mask = nullptr;
  float mslope = 0.f;
  static constexpr bool HAS_VEC4 = true;
  __device__ __forceinline__ void set_batch(int b) {
    y += (int64_t)b * bs;
    if (res) res += (int64_t)b * bs;
    if (mask) mask += (int64_t)b * bs;
  }
  __device__ __forceinline__ void store4(int64_t off, int col, float4 v) const {  // columns col .. col+3 of one row
    if (mask) {
      const float4 m4 = *reinterpret_cast<const float4*>(mask + off + col);
      v.x *= m4.x > 0.f ? 1.f : mslope, v.y *= m4.y > 0.f ? 1.f : mslope, v.z *= m4.z > 0.f ? 1.f : mslope, v.w *= m4.w > 0.f ? 1.f : mslope;
    }
    if (bias) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
      v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
    }
    if (res) {
      const float4 r4 = *reinterpret_cast<const float4*>(res + off + col);
      v.x += r4.x, v.y += r4.y, v.z += r4.z, v.w += r4.w;
    }
    if (act == 1) v.x = tanhf(v.x), v.y = tanhf(v.y), v.z = tanhf(v.z), v.w = tanhf(v.w);
    else if (act == 2) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<float4*>(y + off + col) = v;
  }
  __device__ __forceinline__ int64_t row_off(int row) const {
    const uint32_t n = fdiv((uint32_t)row, dG);
    const uint32_t rem = (uint32_t)row - n * (uint32_t)(GH * GW);
    const uint32_t gy = fdiv(rem, dGW);
    const uint32_t gx = rem - gy * (uint32_t)GW;
    return ((int64_t)((int)n * OHt + (int)gy * OS + py) * OWt + ((int)gx * OS + px)) * cstride;
  }
  __device__ __forceinline__ void store(int64_t off, int col, float v) const {
    if (mask) v *= mask[off + col] > 0.f ? 1.f : mslope;  // linear: a split reduction masks every partial sum
    if (act == 3) {  // split reduction: y was initialised with bias + residual, the partial sums meet through fp32 atomics
      atomicAdd(y + off + col, v);
      return;
    }
    if (bias) v += bias[col];
    if (res) v += res[off + col];
    if (act == 1) v = tanhf(v);
    else if (act == 2) v = fmaxf(v, 0.f);
    y[off + col] = v;
  }
};

struct WgradEp {  // rows = (tap, channel) -> dwf[(wtap*C + c)*K + col], fp32 atomics across the split reduction
  static constexpr bool HAS_VEC4 = false;
  float* dw;
  ConvGeom g;
  int Kout;
  int64_t bs;
  float* dbias;   // target of the extra ones-row (may be null)
  int ones_row;
  __device__ __forceinline__ void set_batch(int b) { dw += (int64_t)b * bs; }
  __device__ __forceinline__ int64_t row_off(int row) const {
    if (row == ones_row) return -1;
    const int t = (int)fdiv((uint32_t)row, g.dC);
    const int i = (int)fdiv((uint32_t)t, g.dntx), j = t - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    return ((int64_t)wtap * g.C + (row - t * g.C)) * Kout;
  }
  __device__ __forceinline__ void store(int64_t off, int col, float v) const {
    if (off < 0) atomicAdd(dbias + col, v);
    else atomicAdd(dw + off + col, v);
  }
};

#ifndef FMI_HOST_EMU
// epilogue shared by the kernels: register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31 of its 32x32 tile
// 4 x 4 transpose inside every quad of lanes (DPP quad_perm): register rho of lane c  <->  register c of lane rho
__device__ __forceinline__ float dpp_quad(float v, bool stage2) {
  const int i = __float_as_int(v);
  return __int_as_float(stage2 ? __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true)    // quad_perm [2,3,0,1]
                               : __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void quad_transpose(float (&a)[4], int c) {
  const bool b0 = c & 1, b1 = c & 2;
#pragma unroll
  for (int p = 0; p < 4; p += 2) {
    const float recv = dpp_quad(b0 ? a[p] : a[p + 1], false);
    a[p] = b0 ? recv : a[p];
    a[p + 1] = b0 ? a[p + 1] : recv;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float recv = dpp_quad(b1 ? a[q] : a[q + 2], true);
    a[q] = b1 ? recv : a[q];
    a[q + 2] = b1 ? a[q + 2] : recv;
  }
}

template <class EP, class T>
__device__ __forceinline__ void store_tile(const EP& ep, f32x16 (&acc)[T::TM][T::TN], int M, int N, int row0, int col0, int lh, int l31) {
  if constexpr (EP::HAS_VEC4) {
    if (ep.vec) {
      // The accumulator layout gives a lane ONE column and 16 rows: 64 four-byte stores per 32 x 32 tile and lane, each wave
      // instruction two 128-byte row pieces.  The texture path is paced per wave instruction, so tiles with a short reduction
      // (ConvTranspose phases, 1x1 convolutions) were bound by their epilogue.  A 4 x 4 transpose inside each quad of lanes turns
      // registers 4g .. 4g+3 (rows 8g + 4 lh + 0..3, column l) into ONE row and four consecutive columns per lane: 16-byte stores
      // (and residual / bias loads), four times fewer instructions.
      const int c = l31 & 3, colq = l31 & ~3;
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + i * 32 + 8 * g + 4 * lh + c;
          const int64_t off = row < M ? ep.row_off(row) : 0;
#pragma unroll
          for (int j = 0; j < T::TN; ++j) {
            float a[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            quad_transpose(a, c);
            const int col = col0 + j * 32 + colq;
            if (row < M && col < N) ep.store4(off, col, make_float4(a[0], a[1], a[2], a[3]));
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= M) continue;
      const int64_t off = ep.row_off(row);
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int col = col0 + j * 32 + l31;
        if (col < N) ep.store(off, col, acc[i][j][r]);
      }
    }
  }
}

// constant 16-byte chunks the LDS-DMA path reads for out-of-range / synthetic operand elements
static __device__ __attribute__((aligned(16))) float fmi_chunk_zero[4] = {0.f, 0.f, 0.f, 0.f};
// A kernel takes the address of the zero chunk ONCE, pinned in scalar registers: written `cond ? p : fmi_chunk_zero` inside a copy loop,
// every use became s_getpc + s_load_dwordx2 from the GOT + s_waitcnt lgkmcnt(0) -- five scalar-memory round trips per reduction tile in
// the issue section of conv3x3_p3_kernel, and (the counter is shared) a wait for every LDS read in flight wherever reads precede it.
__device__ __forceinline__ const float* fmi_zero_chunk_ptr() {
  const float* p = fmi_chunk_zero;
#ifndef FMI_ZCHUNK_NOPIN  // A/B build: the address re-materialised at every use, as before
  asm volatile("" : "+s"(p));
#endif
  return p;
}
static __device__ __attribute__((aligned(16))) float fmi_chunk_one[4] = {1.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ const float* WgradAX::chunk(const DCtx& d, const Tile&) const {
  if (d.k >= g.Mdim()) return nullptr;
  if (d.x == ones_row) return fmi_chunk_one;
  if (d.c.t0 >= g.ntaps()) return nullptr;
  const int n = d.n;
  int iy = d.gy * g.S + d.dy, ix = d.gx * g.S + d.dx;
  if (g.pad_mode) {
    iy = reflect_idx(iy, g.IH);
    ix = reflect_idx(ix, g.IW);
  }
  if ((unsigned)iy >= (unsigned)g.IH || (unsigned)ix >= (unsigned)g.IW) return nullptr;
  return p + ((int64_t)(n * g.IH + iy) * g.IW + ix) * g.cstride + d.c.c0;
}
#endif

#ifndef FMI_HOST_EMU
// =====================================================================================
// The kernel.  grid.x = tiles_m*tiles_n (XCD-remapped), grid.y = batch*ksplit.
// =====================================================================================
template <class LA, class LB, class EP, class T>
__global__ void __launch_bounds__(256) gemm_mfma_f32_kernel(LA la, LB lb, EP ep, int M, int N, int K, int tiles_n,
                                                            int ksplit, int kchunk) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, LDA = T::LDA, LDB = T::LDB;
  constexpr int NLA = BM / 64 > 0 ? BM / 64 : 1, NLB = BN / 64 > 0 ? BN / 64 : 1;  // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (LDA + LDB)];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA;

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

  // per-thread staging coordinates
  int ax[NLA], ak[NLA], bx[NLB], bk[NLB];
  bool aact[NLA], bact[NLB];
  typename LA::Ctx ca[NLA];
  typename LB::Ctx cb[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    if (LA::KMODE) {
      ax[j] = (tid >> 2) + 64 * j;
      ak[j] = (tid & 3) * 4;
      aact[j] = ax[j] < BM;
    } else {
      const int idx = tid + 256 * j;
      ak[j] = idx / (BM / 4);
      ax[j] = (idx % (BM / 4)) * 4;
      aact[j] = ak[j] < BK;
    }
    ca[j] = la.prep(m0 + ax[j]);
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    if (LB::KMODE) {
      bx[j] = (tid >> 2) + 64 * j;
      bk[j] = (tid & 3) * 4;
      bact[j] = bx[j] < BN;
    } else {
      const int idx = tid + 256 * j;
      bk[j] = idx / (BN / 4);
      bx[j] = (idx % (BN / 4)) * 4;
      bact[j] = bk[j] < BK;
    }
    cb[j] = lb.prep(n0 + bx[j]);
  }

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra0[NLA], rb0[NLB];
#if FMI_EXP & 16
  float4 ra1[NLA], rb1[NLB];
#endif
  auto gload = [&](float4 (&ra)[NLA], float4 (&rb)[NLB], int k0) {
#pragma unroll
    for (int j = 0; j < NLA; ++j) ra[j] = aact[j] ? la.load4(ca[j], m0 + ax[j], k0 + ak[j]) : zero4();
#pragma unroll
    for (int j = 0; j < NLB; ++j) rb[j] = bact[j] ? lb.load4(cb[j], n0 + bx[j], k0 + bk[j]) : zero4();
  };
  auto lstore = [&](const float4 (&ra)[NLA], const float4 (&rb)[NLB], int buf) {
    float* a = As + buf * BK * LDA;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (!aact[j]) continue;
      if (LA::KMODE) {
        a[(ak[j] + 0) * LDA + ax[j]] = ra[j].x;
        a[(ak[j] + 1) * LDA + ax[j]] = ra[j].y;
        a[(ak[j] + 2) * LDA + ax[j]] = ra[j].z;
        a[(ak[j] + 3) * LDA + ax[j]] = ra[j].w;
      } else {
        *reinterpret_cast<float4*>(a + ak[j] * LDA + ax[j]) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (!bact[j]) continue;
      if (LB::KMODE) {
        b[(bk[j] + 0) * LDB + bx[j]] = rb[j].x;
        b[(bk[j] + 1) * LDB + bx[j]] = rb[j].y;
        b[(bk[j] + 2) * LDB + bx[j]] = rb[j].z;
        b[(bk[j] + 3) * LDB + bx[j]] = rb[j].w;
      } else {
        *reinterpret_cast<float4*>(b + bk[j] * LDB + bx[j]) = rb[j];
      }
    }
  };
  // one 16-deep tile from LDS buffer `buf`: all fragments first (counted lgkmcnt waits let the MFMAs start as they
  // arrive), then 8 x TM x TN back-to-back MFMAs: the LDS latency is exposed once per tile, not once per k-pair
  auto compute = [&](int buf) {
    const float* a = As + buf * BK * LDA + wm + l31;
    const float* b = Bs + buf * BK * LDB + wn + l31;
#if FMI_X6
    static_assert(BK == 16, "one bf16 k-step per tile");
    {
      float ga[T::TM][8], gb[T::TN][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) ga[i][s] = a[(8 * lh + s) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) gb[j][s] = b[(8 * lh + s) * LDB + j * 32];
      }
      bf16x8_t pa[T::TM][3], pb[T::TN][3];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) split3_bf16(ga[i], pa[i]);
#pragma unroll
      for (int j = 0; j < T::TN; ++j) split3_bf16(gb[j], pb[j]);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = mfma_x6(pa[i], pb[j], acc[i][j]);
      return;
    }
#endif
    float fa[BK / 2][T::TM], fb[BK / 2][T::TN];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i) fa[kk][i] = a[(2 * kk + lh) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < T::TN; ++j) fb[kk][j] = b[(2 * kk + lh) * LDB + j * 32];
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    }
  };

  if (k_begin < k_end) {
    gload(ra0, rb0, k_begin);
    lstore(ra0, rb0, 0);
  }
#if FMI_EXP & 16
  // two register sets: the loads of tile t+2 are issued before tile t is computed (prefetch distance 2)
  if (k_begin + BK < k_end) gload(ra1, rb1, k_begin + BK);
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
    if (k0 + 2 * BK < k_end) gload(ra0, rb0, k0 + 2 * BK);
    compute(buf);
    if (k0 + BK < k_end) lstore(ra1, rb1, buf ^ 1);
    __syncthreads();
    buf ^= 1;
    if (k0 + BK >= k_end) break;
    if (k0 + 3 * BK < k_end) gload(ra1, rb1, k0 + 3 * BK);
    compute(buf);
    if (k0 + 2 * BK < k_end) lstore(ra0, rb0, buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#else
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const bool more = k0 + BK < k_end;
#if !(FMI_EXP & 2)
    if (more) gload(ra0, rb0, (FMI_EXP & 8) ? k_begin : k0 + BK);
#endif
    compute(buf);
#if !(FMI_EXP & 4)
    if (more) lstore(ra0, rb0, buf ^ 1);
#endif
#if !(FMI_EXP & 1)
    __syncthreads();
#endif
    buf ^= 1;
  }
#endif

  store_tile<EP, T>(ep, acc, M, N, m0 + wm, n0 + wn, lh, l31);
}

// =====================================================================================
// LDS-DMA pipeline (the default whenever both operands can be fetched in aligned 16-byte chunks).
//
// Operand tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of
// FMI_NST 16-deep stages: the copies of tile t+FMI_NST-1 are issued right after the barrier that opens tile t, so a copy
// has a whole MFMA phase (~2048 cycles per wave, x the 3 workgroups that share a CU) to land.  The register-staged kernel
// above exposes about half of the global-load latency and pays ~15 % in staging instructions (ablation table in
// DESIGN.md).  Measured: 2 stages = 3 stages (125 vs 126 TFLOP/s at 4096^3) and better on narrow tiles (less LDS, more
// workgroups per CU); 4 stages lose 5 % to occupancy.
//
// LDS image of a stage (a DMA instruction writes 64 consecutive 16-byte chunks, lane-linear):
//   reduction-contiguous operand (KMODE): [row][16 k] ; chunk position (row, q ^ ((row >> 2) & 3)) holds k-quarter q: the
//       XOR goes on the SOURCE address of the copy and on the reader's address; ds_read_b128 fragment reads are
//       conflict-free in the hardware's 16-lane groups.
//   row-contiguous operand: [16 k][rows]; ds_read_b32 fragment reads of 32 consecutive rows per half-wave.
// MFMA k assignment inside a tile: half-wave h (= lane >> 5) owns k = 8h .. 8h+7, step s multiplies A[:, 8h+s] B[8h+s, :]
// (any pairing is legal as long as both operands use the same one).
// Synchronisation: one raw s_barrier per tile.  Before it every wave waits (counted vmcnt) for its own copies of tile t;
// after it the stage read one tile ago is free (all waves finished its ds_reads before arriving) and is refilled.
// =====================================================================================
#ifndef FMI_DMA_ATTR
#define FMI_DMA_ATTR
#endif
template <class L, class = void>
struct is_split3 : std::false_type {};
template <class L>
struct is_split3<L, std::void_t<decltype(L::SPLIT3)>> : std::integral_constant<bool, L::SPLIT3> {};

// FL: blocked accumulation -- a second accumulator set, flushed every 32 tiles (512 reduction elements), so that a long unsplit
// reduction is a sum of short fp32 chains (conv_p3.h)
template <class LA, class LB, class EP, class T, bool FL = false>
__global__ void __launch_bounds__(256) FMI_DMA_ATTR gemm_dma_f32_kernel(LA la, LB lb, EP ep, int M, int N, int K, int tiles_n,
                                                           int ksplit, int kchunk) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  #ifndef FMI_NST
#define FMI_NST 2
#endif
  constexpr int BM = T::BM, BN = T::BN, BK = 16, NST = FMI_NST, DEPTH = NST - 1;  // DEPTH tiles are copied ahead of the one computed
  // copies: a tile image has BX*4 16-byte chunks = BX/16 wave instructions; wave w issues instructions w, w+4, ...
  constexpr bool B3 = is_split3<LB>::value;  // B arrives as three bf16 piece images [2 channel groups][BN] x 16 bytes
  constexpr int NIB3 = 3 * BN / 32;         // wave instructions of the three piece images of a tile
  constexpr int NLA = (BM + 63) / 64, NLB = B3 ? (NIB3 + 3) / 4 : (BN + 63) / 64;
  constexpr int STAGE = BM * BK + (B3 ? 3 * BN * 8 : BN * BK);  // floats
  __shared__ __attribute__((aligned(1024))) float lds[NST * STAGE];

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

  // which chunk of the tile each of this thread's copies fetches (position p = j*256 + tid of the lane-linear image)
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
    if constexpr (B3) {  // wave instruction j * 4 + wid of [3 pieces][2 channel groups][BN] chunks
      const int p = (j * 4 + wid) * 64 + lane;
      const int piece = p / (2 * BN), r = p - piece * (2 * BN), kg = r / BN, x = r - kg * BN;
      db[j] = lb.dprep3(piece < 3 ? piece : 0, kg, piece < 3 ? n0 + x : 0x40000000);
    } else {
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
  }
  // wave-uniform: does copy slot j of this wave exist (tiles narrower than 64 rows fill only waves 0..1)
  const int na_w = (BM % 64 == 0) ? NLA : (wid * 64 < BM * 4 ? 1 : 0);
  const int nb_w = B3 ? (NIB3 - wid + 3) / 4 : ((BN % 64 == 0) ? NLB : (wid * 64 < BN * 4 ? 1 : 0));

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // The copies are issued from inline asm: hipcc then neither counts them nor drains them (it would put an
  // s_waitcnt vmcnt(0) in front of every ds_read that follows a compiler-visible LDS-DMA); their completion is counted
  // by hand below.  M0 carries the wave-uniform LDS byte address; lane l writes bytes [16 l, 16 l + 16) after it.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto glds16 = [&](const float* g, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
  };
  auto issue = [&](int k0, int st) {
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE + wid * 256) * 4u);  // this wave's 1 KiB slice
    const uint32_t sb = sa + BM * BK * 4;
    const typename LA::Tile ta = la.tile(k0);
    const typename LB::Tile tb = lb.tile(k0);
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (BM % 64 != 0 && !na_w) break;
      const float* g = la.chunk(da[j], ta);
      if (!g) g = zchunk;
      glds16(g, sa + j * 4096);
      la.advance(da[j]);
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if constexpr (B3) {
        if (j >= nb_w) break;
        const void* g = lb.chunk(db[j], tb);
        if (!g) g = zchunk;
        glds16((const float*)g, sb + j * 4096);
      } else {
        if (BN % 64 != 0 && !nb_w) break;
        const float* g = lb.chunk(db[j], tb);
        if (!g) g = zchunk;
        glds16(g, sb + j * 4096);
        lb.advance(db[j]);
      }
    }
  };
  auto compute = [&](int st) {
    const float* sa = lds + st * STAGE;
    const float* sb = sa + BM * BK;
    float fa[T::TM][8], fb[T::TN][8];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      if (LA::KMODE) {
        const int r = wm + i * 32 + l31, sw = (r >> 2) & 3;
        const float4 v0 = *reinterpret_cast<const float4*>(sa + r * 16 + ((2 * lh) ^ sw) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(sa + r * 16 + ((2 * lh + 1) ^ sw) * 4);
        fa[i][0] = v0.x, fa[i][1] = v0.y, fa[i][2] = v0.z, fa[i][3] = v0.w;
        fa[i][4] = v1.x, fa[i][5] = v1.y, fa[i][6] = v1.z, fa[i][7] = v1.w;
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) fa[i][s] = sa[(8 * lh + s) * BM + wm + i * 32 + l31];
      }
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      if constexpr (B3) {
        break;  // the piece fragments are read below
      } else if (LB::KMODE) {
        const int r = wn + j * 32 + l31, sw = (r >> 2) & 3;
        const float4 v0 = *reinterpret_cast<const float4*>(sb + r * 16 + ((2 * lh) ^ sw) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(sb + r * 16 + ((2 * lh + 1) ^ sw) * 4);
        fb[j][0] = v0.x, fb[j][1] = v0.y, fb[j][2] = v0.z, fb[j][3] = v0.w;
        fb[j][4] = v1.x, fb[j][5] = v1.y, fb[j][6] = v1.z, fb[j][7] = v1.w;
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) fb[j][s] = sb[(8 * lh + s) * BN + wn + j * 32 + l31];
      }
    }
#if FMI_X6
    bf16x8_t pa[T::TM][3], pb[T::TN][3];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) split3_bf16(fa[i], pa[i]);
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      if constexpr (B3) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          pb[j][pc] = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const unsigned char*>(sb) + pc * (2 * BN * 16) + (lh * BN + wn + j * 32 + l31) * 16);
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
  };

  auto wait_copies = [&](int n) {  // s_waitcnt vmcnt(n): at most n of this wave's copies may still be in flight
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
  };
  const int nt = (k_end - k_begin + BK - 1) / BK;
  const int nw = (!B3 && BM % 64 == 0 && BN % 64 == 0) ? NLA + NLB : na_w + nb_w;  // copies this wave issues per tile
#pragma unroll
  for (int p = 0; p < DEPTH; ++p)
    if (p < nt) issue(k_begin + p * BK, p);
  int st = 0, stn = DEPTH;  // stage of tile t, stage tile t+DEPTH goes to
  f32x16 acc2[FL ? T::TM : 1][FL ? T::TN : 1];
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  }
  for (int t = 0; t < nt; ++t) {
    // tile t must have landed; the newer tiles already issued (at most DEPTH-1 of them) may stay in flight
    int pend = nt - 1 - t;
    if (pend > DEPTH - 1) pend = DEPTH - 1;
    wait_copies(pend * nw);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + DEPTH < nt) issue(k_begin + (t + DEPTH) * BK, stn);
#if FMI_EXP & 64
    __builtin_amdgcn_s_setprio(1);
#endif
    compute(st);
#if FMI_EXP & 64
    __builtin_amdgcn_s_setprio(0);
#endif
    st = st == NST - 1 ? 0 : st + 1;
    stn = stn == NST - 1 ? 0 : stn + 1;
    if constexpr (FL) {
      if ((t & 31) == 31 || t + 1 == nt) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              acc2[i][j][r] += acc[i][j][r];
              acc[i][j][r] = 0.f;
            }
      }
    }
  }
  if constexpr (FL) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = acc2[i][j];
  }

  store_tile<EP, T>(ep, acc, M, N, m0 + wm, n0 + wn, lh, l31);
}

// Host-side launcher: picks the tile from N (and M), validates the 32-bit index ranges the kernel assumes.
template <class LA, class LB, class EP>
static int launch_gemm(const LA& la, const LB& lb, const EP& ep, int M, int N, int K, int batch, int ksplit,
                       hipStream_t st) {
  if (M <= 0 || N <= 0 || K < 0 || batch <= 0 || ksplit <= 0) return FMI_ERR_BAD_ARG;
  int kchunk = 16;
  if (K == 0) {
    ksplit = 1;  // empty reduction: the epilogue still writes bias / residual
  } else {
    kchunk = (int)ceil_div64(ceil_div64(K, ksplit), 16) * 16;
    ksplit = (int)ceil_div64(K, kchunk);
  }
  const int64_t gy = (int64_t)batch * ksplit;
  if (gy > 65535) return FMI_ERR_UNSUPPORTED;
  // FMI_DMA_OFF (debug): bit 0 dense GEMM, bit 1 conv forward / adjoint, bit 2 weight gradient -> use the register-staged kernel
  static const int dma_off = getenv("FMI_DMA_OFF") ? atoi(getenv("FMI_DMA_OFF")) : 0;
  const int family = std::is_same<LA, ConvK>::value ? 2 : (std::is_same<LA, WgradAX>::value ? 4 : 1);
  bool dma = !(FMI_EXP & 32) && !(dma_off & family) && la.dma_ok() && lb.dma_ok();
  if (const char* r = getenv("FMI_DMA_OFF_RANGE")) {  // debug: "a:b" = launches a <= i < b of this process use the register-staged kernel
    long a = 0, b = 0;
    sscanf(r, "%ld:%ld", &a, &b);
    const long i = __atomic_fetch_add(&fmi_debug_launch_counter, 1L, __ATOMIC_RELAXED);  // forward and autograd threads both launch
    if (i >= a && i < b) dma = false;
    if (getenv("FMI_DMA_TRACE")) fprintf(stderr, "[fmi launch %ld] family %d M %d N %d K %d batch %d ksplit %d dma %d (eligible %d %d)\n", i, family, M, N, K, batch, ksplit, (int)dma, (int)la.dma_ok(), (int)lb.dma_ok());
  }
  const bool fl = kchunk > 640 && (fmi_det() || fmi_blocked_acc());  // blocked accumulation of a long unsplit reduction (LDS-DMA kernel)
#define FMI_LAUNCH(TILE)                                                                                      \
  do {                                                                                                        \
    const int64_t tm = ceil_div64(M, TILE::BM), tn = ceil_div64(N, TILE::BN);                                 \
    if (tm * tn > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;                                                   \
    {                                                                                                         \
      if (dma && fl) {                                                                                        \
        hipLaunchKernelGGL((gemm_dma_f32_kernel<LA, LB, EP, TILE, true>), dim3((unsigned)(tm * tn), (unsigned)gy), dim3(256), \
                           0, st, la, lb, ep, M, N, K, (int)tn, ksplit, kchunk);                              \
        break;                                                                                                \
      }                                                                                                       \
      if (dma) {                                                                                              \
        hipLaunchKernelGGL((gemm_dma_f32_kernel<LA, LB, EP, TILE>), dim3((unsigned)(tm * tn), (unsigned)gy), dim3(256), \
                           0, st, la, lb, ep, M, N, K, (int)tn, ksplit, kchunk);                              \
        break;                                                                                                \
      }                                                                                                       \
    }                                                                                                         \
    hipLaunchKernelGGL((gemm_mfma_f32_kernel<LA, LB, EP, TILE>), dim3((unsigned)(tm * tn), (unsigned)gy), dim3(256), \
                       0, st, la, lb, ep, M, N, K, (int)tn, ksplit, kchunk);                                  \
  } while (0)
  // largest tile that still gives ~1.5 workgroups per CU; small problems trade operand reuse for occupancy
  const int64_t zs = gy;
  auto wgs = [&](int bm, int bn) { return ceil_div64(M, bm) * ceil_div64(N, bn) * zs; };
  const int64_t want = 384, want128 = 800;  // 128x128 only from ~one full round of workgroups (3 per CU) up: below that the
                                           // fullest CUs set the time and the 64x128 tile balances better (measured: -0.7 ms per step)
  if (N <= 32) {
    FMI_LAUNCH(Tile128x32);
  } else if (N <= 64) {
    if (M > 64 && wgs(128, 64) >= want) FMI_LAUNCH(Tile128x64);
    else FMI_LAUNCH(Tile64x64);
  } else {
    if (M > 64 && wgs(128, 128) >= want128) FMI_LAUNCH(Tile128x128);
    else if (M > 32 && wgs(64, 128) >= want) FMI_LAUNCH(Tile64x128);
    else if (M > 64 && wgs(32, 128) < 256) FMI_LAUNCH(Tile64x128);
    else FMI_LAUNCH(Tile32x128);
  }
#undef FMI_LAUNCH
  return fmi_launch_status();
}

#else  // ------------------------------ FMI_HOST_EMU ------------------------------
// Reference evaluation of the same (loader, loader, epilogue) triple with plain loops.
template <class L>
static void emu_dense(const L& l, int X, int K, float* out /*[X][K]*/) {
  if (L::KMODE) {
    for (int x = 0; x < X; ++x) {
      auto c = l.prep(x);
      for (int k = 0; k < ((K + 3) / 4) * 4; k += 4) {
        float4 v = l.load4(c, x, k);
        const float e[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; ++i)
          if (k + i < K) out[(int64_t)x * K + k + i] = e[i];
      }
    }
  } else {
    for (int x = 0; x < ((X + 3) / 4) * 4; x += 4) {
      auto c = l.prep(x);
      for (int k = 0; k < K; ++k) {
        float4 v = l.load4(c, x, k);
        const float e[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; ++i)
          if (x + i < X) out[(int64_t)(x + i) * K + k] = e[i];
      }
    }
  }
}
template <class LA, class LB, class EP>
static int launch_gemm(const LA& la0, const LB& lb0, const EP& ep0, int M, int N, int K, int batch, int ksplit,
                       hipStream_t) {
  if (M <= 0 || N <= 0 || K < 0 || batch <= 0 || ksplit <= 0) return FMI_ERR_BAD_ARG;
  float* A = new float[(int64_t)M * (K + 1)];
  float* B = new float[(int64_t)N * (K + 1)];
  for (int b = 0; b < batch; ++b) {
    LA la = la0;
    LB lb = lb0;
    EP ep = ep0;
    la.set_batch(b);
    lb.set_batch(b);
    ep.set_batch(b);
    memset(A, 0, sizeof(float) * (int64_t)M * (K + 1));
    memset(B, 0, sizeof(float) * (int64_t)N * (K + 1));
    emu_dense(la, M, K, A);
    emu_dense(lb, N, K, B);
    for (int m = 0; m < M; ++m) {
      const int64_t off = ep.row_off(m);
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)A[(int64_t)m * K + k] * B[(int64_t)n * K + k];
        ep.store(off, n, (float)acc);
      }
    }
  }
  delete[] A;
  delete[] B;
  return FMI_OK;
}
#endif
