// bf16 implicit-GEMM convolution family for the StyleGAN2 decoder of the pSp path (stylegan2/model.py:241-279, configs C3 / C5:
// bf16 compute, fp32 accumulate): forward, adjoint (= ConvTranspose2d forward / input gradient) and weight gradient.
//
// Machine mapping (gfx950):
//  * v_mfma_f32_32x32x16_bf16: lane l supplies A[row l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][col l&31] as eight bf16
//    (one 16-byte register quad); the 32x32 fp32 result has the layout of the fp32 core (gemm_core.h).
//  * operands reach LDS by global_load_lds_dwordx4 (LDS-DMA), 16 bytes = 8 bf16 per lane, two-stage ring, one raw s_barrier and a
//    counted s_waitcnt vmcnt per reduction tile -- the pipeline of gemm_dma_f32_kernel.
//  * forward / adjoint: both operands have the reduction index contiguous in memory (NHWC pixels x (tap, channel); weights packed
//    [out][tap][in]), so a tile image is [row][BK bf16] and ONE ds_read_b128 is one MFMA operand.  BK = 64 (128-byte rows, whole
//    cache lines per row) when the channel count allows, else 32.  The 16-byte chunk index is XOR-swizzled on the SOURCE address
//    (the DMA writes lane-linear) and on the read: (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte rows -- every 16-lane
//    group of a ds_read_b128 then covers all 64 banks once.
//  * weight gradient: the reduction runs over pixels, which are the SLOW index of both x[pixel][c] and dy[pixel][k]; the MFMA
//    operands come from ds_read_b64_tr_b16 (hardware transpose of a 4-pixel x 16-row block per 16 lanes), two per operand.  The
//    image is [32-row group][64 pixels][32 rows]: the four pixel rows one transposed read touches are 256 contiguous bytes (one
//    bank row, conflict-free without a swizzle) and all copies of a thread belong to one pixel (one anchor decode per tile).
#include "gemm_core.h"

#ifndef FMI_HOST_EMU
typedef uint16_t bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint16_t f2bf(float f) {  // round to nearest even (inputs are finite)
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static bool aligned16b(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------------------
// loaders (LDS-DMA hooks only: chunk = 8 consecutive reduction elements of one row)
// ---------------------------------------------------------------------------------------------------------------------------
struct ConvKB {  // A operand: gathered pixels x (tap, channel); a reduction tile lies inside one tap (C % BK == 0), zero padding
  const bf16_t* p;
  ConvGeom g;
  struct DCtx {
    int64_t boff;
    int ry, rx;
  };
  struct Tile {
    int dy, dx;
    int64_t uoff;
  };
  __device__ DCtx dprep(int x, int kq) const {
    DCtx d;
    if (x >= g.Mdim()) {
      d.ry = -0x20000000, d.rx = 0, d.boff = 0;
      return d;
    }
    const uint32_t n = fdiv((uint32_t)x, g.dG);
    const uint32_t rem = (uint32_t)x - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    d.ry = (int)gy * g.S + g.dy0;
    d.rx = (int)gx * g.S + g.dx0;
    d.boff = ((int64_t)((int)n * g.IH + d.ry) * g.IW + d.rx) * g.cstride + kq;
    return d;
  }
  // Reduction order: channel chunk (one reduction tile of BK = 1 << g.vec channels) OUTER, tap INNER -- the taps of a chunk re-read
  // the same pixels, so they follow each other while those bytes are in L2 (tap-major order re-fetched the activation once per tap:
  // FETCH_SIZE 4x the operand bytes at 512 channels).  g.dC divides by the number of taps here.
  __device__ Tile tile(int k0) const {
    const int kt = k0 >> g.vec;
    const int chunk = (int)fdiv((uint32_t)kt, g.dC), tp = kt - chunk * g.ntaps();
    const int i = (int)fdiv((uint32_t)tp, g.dntx), j = tp - i * g.ntx;
    Tile t;
    t.dy = g.ystep * i;
    t.dx = g.xstep * j;
    t.uoff = ((int64_t)(g.ystep * i) * g.IW + t.dx) * g.cstride + (chunk << g.vec);
    return t;
  }
  __device__ const bf16_t* chunk(const DCtx& d, const Tile& t) const {
    const bool ok = (unsigned)(d.ry + t.dy) < (unsigned)g.IH && (unsigned)(d.rx + t.dx) < (unsigned)g.IW;
    return ok ? p + d.boff + t.uoff : nullptr;
  }
};

struct ConvWKB {  // B operand: weights packed [Nout][T taps][Cred] (reduction contiguous)
  const bf16_t* p;
  ConvGeom g;  // tap algebra and C
  int Nout, T;
  struct DCtx {
    int64_t off;
  };
  struct Tile {
    int toff;
  };
  __device__ DCtx dprep(int x, int kq) const { return DCtx{x < Nout ? (int64_t)x * T * g.C + kq : -1}; }
  __device__ Tile tile(int k0) const {
    const int kt = k0 >> g.vec;
    const int chunk = (int)fdiv((uint32_t)kt, g.dC), tp = kt - chunk * g.ntaps();
    const int i = (int)fdiv((uint32_t)tp, g.dntx), j = tp - i * g.ntx;
    const int wtap = (g.kh0 + g.khstep * i) * g.kw + (g.kw0 + g.kwstep * j);
    return Tile{wtap * g.C + (chunk << g.vec)};
  }
  __device__ const bf16_t* chunk(const DCtx& d, const Tile& t) const { return (d.off >= 0 && t.toff >= 0) ? p + d.off + t.toff : nullptr; }
};

struct ConvEpB {  // rows = anchors, written at (gy*OS+py, gx*OS+px) of the bf16 tensor [N][OHt][OWt][cstride]
  bf16_t* y;
  const float* colscale;  // [N][Nout] or null: y *= colscale[sample][col] (demodulation, model.py:250-252)
  int GH, GW, OS, py, px, OHt, OWt, cstride, Nout;
  FastDiv dGW, dG;
  int vec;
  __device__ int64_t row_pix(int row, int& n_out) const {  // pixel index of the output tensor
    const uint32_t n = fdiv((uint32_t)row, dG);
    const uint32_t rem = (uint32_t)row - n * (uint32_t)(GH * GW);
    const uint32_t gy = fdiv(rem, dGW);
    const uint32_t gx = rem - gy * (uint32_t)GW;
    n_out = (int)n;
    return (int64_t)((int)n * OHt + (int)gy * OS + py) * OWt + ((int)gx * OS + px);
  }
  __device__ int64_t row_off(int row, int& n_out) const { return row_pix(row, n_out) * cstride; }
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward / adjoint kernel
// ---------------------------------------------------------------------------------------------------------------------------
// tiles of the bf16 kernels: WM x WN waves (4 or 8), each TM x TN accumulators of 32 x 32
template <int WM_, int WN_, int TM_, int TN_>
struct TileB {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
};
using TB128x128 = TileB<2, 2, 2, 2>;
using TB128x64 = TileB<2, 2, 2, 1>;
using TB128x32 = TileB<4, 1, 1, 1>;
using TB64x64 = TileB<2, 2, 1, 1>;
using TB64x128 = TileB<2, 2, 1, 2>;
using TB256x256 = TileB<2, 4, 4, 2>;  // 8 waves, 128 x 64 per wave: 128 FLOP per staged byte (the 128 x 128 tile: 64 -> L2-rate bound)

// One launch = up to four "phases" (the sub-pixel phases of a strided adjoint, i.e. of a ConvTranspose2d forward; one phase
// otherwise) x an optional split of the reduction: grid = (tiles of the largest phase, phases, splits).
struct ConvPhaseB {
  ConvKB la;
  ConvWKB lb;
  ConvEpB ep;
  int M, K, tiles;
};
struct ConvSetB {
  ConvPhaseB ph[4];
  int N, tiles_n, ksplit;
  float* ws;  // split reduction: fp32 twin of the output (zeroed by the caller), partial sums meet through atomics
};

template <class T, int BK, int NST = 2>
__global__ void __launch_bounds__(T::NT) conv_bf16_kernel(ConvSetB set) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  if (lid >= set.ph[blockIdx.y].tiles) return;  // whole workgroup
  const ConvKB la = set.ph[blockIdx.y].la;
  const ConvWKB lb = set.ph[blockIdx.y].lb;
  const ConvEpB ep = set.ph[blockIdx.y].ep;
  const int M = set.ph[blockIdx.y].M, K = set.ph[blockIdx.y].K, N = set.N, tiles_n = set.tiles_n;
  constexpr int CPR = BK / 8;            // 16-byte chunks per image row
  constexpr int NLA = (BM * CPR + NT - 1) / NT, NLB = (BN * CPR + NT - 1) / NT;
  constexpr int STAGE = (BM + BN) * BK;  // bf16 elements
  constexpr int DEPTH = NST - 1;  // tiles copied ahead of the one computed
  __shared__ __attribute__((aligned(1024))) bf16_t lds[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = (wid / T::WN) * T::TM * 32, wn = (wid % T::WN) * T::TN * 32;

  auto swz = [](int row) { return CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3; };

  ConvKB::DCtx da[NLA];
  ConvWKB::DCtx db[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int p = j * NT + tid, x = p / CPR, c = (p % CPR) ^ swz(x);
    da[j] = la.dprep(m0 + x, c * 8);
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int p = j * NT + tid, x = p / CPR, c = (p % CPR) ^ swz(x);
    db[j] = lb.dprep(n0 + x, c * 8);
  }
  // images smaller than 256 chunks are filled by the first waves only (wave-uniform)
  static_assert(BM * CPR % NT == 0 || BM * CPR < NT, "A image: whole instructions per thread, or fewer chunks than threads");
  static_assert(BN * CPR % NT == 0 || BN * CPR < NT, "B image: whole instructions per thread, or fewer chunks than threads");
  const int na_w = (BM * CPR % NT == 0) ? NLA : (wid * 64 < BM * CPR ? 1 : 0);
  const int nb_w = (BN * CPR % NT == 0) ? NLB : (wid * 64 < BN * CPR ? 1 : 0);

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)lds;
  auto glds16 = [&](const void* g, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
  };
  auto issue = [&](int k0, int st) {
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(st * STAGE) * 2u + (uint32_t)wid * 1024u);
    const uint32_t sb = sa + BM * BK * 2;
    const ConvKB::Tile ta = la.tile(k0);
    const ConvWKB::Tile tb = lb.tile(k0);
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      if (BM * CPR % NT != 0 && !na_w) break;
      const void* g = la.chunk(da[j], ta);
      if (!g) g = zchunk;
      glds16(g, sa + j * NT * 16);
    }
#pragma unroll
    for (int j = 0; j < NLB; ++j) {
      if (BN * CPR % NT != 0 && !nb_w) break;
      const void* g = lb.chunk(db[j], tb);
      if (!g) g = zchunk;
      glds16(g, sb + j * NT * 16);
    }
  };
  auto compute = [&](int st) {
    const bf16_t* sa = lds + st * STAGE;
    const bf16_t* sb = sa + BM * BK;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 fa[T::TM], fb[T::TN];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const int r = wm + i * 32 + l31;
        fa[i] = *reinterpret_cast<const bf16x8*>(sa + r * BK + (((2 * s + lh) ^ swz(r)) << 3));
      }
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        const int r = wn + j * 32 + l31;
        fb[j] = *reinterpret_cast<const bf16x8*>(sb + r * BK + (((2 * s + lh) ^ swz(r)) << 3));
      }
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  auto wait_copies = [&](int n) {
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
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  const int nt_all = (K + BK - 1) / BK, per = (nt_all + set.ksplit - 1) / set.ksplit;
  const int kt0 = blockIdx.z * per;
  const int nt = (kt0 + per <= nt_all ? per : nt_all - kt0);  // reduction tiles of this split (may be <= 0)
  // no reduction tile for this split: nothing to add -- EXCEPT a phase with an empty reduction (K = 0: the three tap-less sub-pixel
  // phases of the adjoint of a 1x1 stride-2 convolution, the IR-SE shortcut of the pSp encoder), whose outputs are zeros that must be
  // WRITTEN when the result goes straight to y (with a zero-initialised split workspace they already are)
  if (nt <= 0 && (K > 0 || blockIdx.z > 0 || set.ws)) return;
  const int nw = (BM * CPR % NT == 0 && BN * CPR % NT == 0) ? NLA + NLB : na_w + nb_w;  // copies this wave issues per tile
#pragma unroll
  for (int p = 0; p < DEPTH; ++p)
    if (p < nt) issue((kt0 + p) * BK, p);
  int st = 0, stn = DEPTH;
  for (int t = 0; t < nt; ++t) {
    int pend = nt - 1 - t;  // tile t must have landed; up to DEPTH-1 newer tiles may stay in flight
    if (pend > DEPTH - 1) pend = DEPTH - 1;
    wait_copies(pend * nw);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + DEPTH < nt) issue((kt0 + t + DEPTH) * BK, stn);
    compute(st);
    st = st == NST - 1 ? 0 : st + 1;
    stn = stn == NST - 1 ? 0 : stn + 1;
  }

  // epilogue: 4 x 4 transpose inside each quad of lanes -> one row, four consecutive columns per lane, 8-byte bf16 stores
  const int c = l31 & 3, colq = l31 & ~3;
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = m0 + wm + i * 32 + 8 * g + 4 * lh + c;
      int n_s = 0;
      const int64_t off = row < M ? ep.row_off(row, n_s) : 0;
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
        float a[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        quad_transpose(a, c);
        const int col = n0 + wn + j * 32 + colq;
        if (row >= M || col >= N) continue;
        if (set.ws) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) atomicAdd(set.ws + off + col + e, a[e]);
          continue;
        }
        if (ep.colscale) {
          const float* cs = ep.colscale + (int64_t)n_s * ep.Nout + col;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) a[e] *= cs[e];
        }
        if (ep.vec) {
          uint2 v;
          v.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16);
          v.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
          *reinterpret_cast<uint2*>(ep.y + off + col) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) ep.y[off + col + e] = f2bf(a[e]);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    uint2 o;
    o.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    o.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    reinterpret_cast<uint2*>(dst)[i] = o;
  }
}

#include "conv_bf16_8ph.h"

// kernel selection override (tests / A-B timing): 0 = by shape, 1 = never the 8-wave tiles, 2 = no eight-phase kernel, 8 = the
// eight-phase kernel wherever it is legal.  FMI_BF16_TILE in the environment sets the initial value.
static int g_bf16_tile_mode = -1;
static int bf16_tile_mode() {
  if (g_bf16_tile_mode < 0) g_bf16_tile_mode = getenv("FMI_BF16_TILE") ? atoi(getenv("FMI_BF16_TILE")) : 0;
  return g_bf16_tile_mode;
}
extern "C" int fmi_debug_bf16_tile(int mode) {
  const int prev = bf16_tile_mode();
  if (mode >= 0) g_bf16_tile_mode = mode;
  return prev;
}

// phases: nph filled entries of set.ph (la, lb, ep, M, K); N = output channels.  ws (may be null): zeroed fp32 twin of the output
// tensor with out_elems elements -- small feature maps with a deep reduction (the 4^2 .. 16^2 layers: 2 - 32 output tiles) then
// split the reduction over workgroups and the sums are converted to bf16 by a second launch.
template <int BK>
static int launch_conv_bf16(ConvSetB& set, int nph, int N, bf16_t* y, float* ws, int64_t ws_floats, int64_t out_elems, hipStream_t st,
                            const EpActB* act = nullptr) {
  int Mmax = 0, Kmax = 0;
  for (int p = 0; p < nph; ++p) {
    if (set.ph[p].M > Mmax) Mmax = set.ph[p].M;
    if (set.ph[p].K > Kmax) Kmax = set.ph[p].K;
  }
  set.N = N;
  const bool can_split = ws && ws_floats >= out_elems && out_elems % 4 == 0 && !set.ph[0].ep.colscale && !fmi_det();  // reproducible mode: no split
#define FMI_LAUNCH_B(TILE)                                                                                                        \
  do {                                                                                                                            \
    const int64_t tn = ceil_div64(N, TILE::BN);                                                                                   \
    int64_t tmax = 0, tsum = 0;                                                                                                   \
    for (int p = 0; p < nph; ++p) {                                                                                               \
      const int64_t t = ceil_div64(set.ph[p].M, TILE::BM) * tn;                                                                   \
      if (t > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;                                                                           \
      set.ph[p].tiles = (int)t;                                                                                                   \
      tsum += t;                                                                                                                  \
      if (t > tmax) tmax = t;                                                                                                     \
    }                                                                                                                             \
    int64_t ks = 1;                                                                                                               \
    if (can_split && tsum < 128 && Kmax >= 1024) {                                                                                \
      ks = ceil_div64(256, tsum);                                                                                                 \
      if (ks > Kmax / 256) ks = Kmax / 256;                                                                                       \
    }                                                                                                                             \
    set.tiles_n = (int)tn;                                                                                                        \
    set.ksplit = (int)ks;                                                                                                         \
    set.ws = ks > 1 ? ws : nullptr;                                                                                               \
    hipLaunchKernelGGL((conv_bf16_kernel<TILE, BK>), dim3((unsigned)tmax, nph, (unsigned)ks), dim3(TILE::NT), 0, st, set);        \
    if (ks > 1)                                                                                                                   \
      hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(fmi_bw_grid(out_elems / 4, 256)), dim3(256), 0, st, ws, y, out_elems / 4);      \
  } while (0)
  auto wgs = [&](int bm, int bn) { return ceil_div64(Mmax, bm) * ceil_div64(N, bn) * nph; };
  const int tile_dbg = bf16_tile_mode();
  // the eight-phase kernel (conv_bf16_8ph.h): whole 64-deep reduction tiles, 8-byte output stores, no split reduction
  bool ok8 = BK == 64 && tile_dbg != 1 && tile_dbg != 2 && N % 4 == 0 && N > 64;
  for (int p = 0; p < nph && ok8; ++p)
    ok8 = set.ph[p].K % 64 == 0 && set.ph[p].la.g.ntaps() <= 32 && set.ph[p].ep.vec && (!set.ph[p].ep.colscale || (((uintptr_t)set.ph[p].ep.colscale & 15) == 0));
  if (act && act->on && !ok8) return FMI_ERR_UNSUPPORTED;
  if (ok8) {
    const bool wide = N > 128;
    const int64_t w8 = wide ? wgs(256, 256) : wgs(512, 128);
    if (tile_dbg == 8 || (act && act->on) || w8 >= 200) {
      const int64_t tn = ceil_div64(N, wide ? 256 : 128);
      int64_t tmax = 0;
      for (int p = 0; p < nph; ++p) {
        const int64_t t = ceil_div64(set.ph[p].M, wide ? 256 : 512) * tn;
        if (t > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
        set.ph[p].tiles = (int)t;
        if (t > tmax) tmax = t;
      }
      set.tiles_n = (int)tn;
      set.ksplit = 1;
      set.ws = nullptr;
      EpActB a{};
      if (act) a = *act;
      if (wide) hipLaunchKernelGGL((conv_bf16_8ph_kernel<T8P256x256>), dim3((unsigned)tmax, nph, 1), dim3(512), 0, st, set, a);
      else hipLaunchKernelGGL((conv_bf16_8ph_kernel<T8P512x128>), dim3((unsigned)tmax, nph, 1), dim3(512), 0, st, set, a);
      return fmi_launch_status();
    }
  }
  if (N <= 32) {
    FMI_LAUNCH_B(TB128x32);
  } else if (N <= 64) {
    if (wgs(128, 64) >= 384) FMI_LAUNCH_B(TB128x64);
    else FMI_LAUNCH_B(TB64x64);
  } else {
    // measured (512 -> 512 at 64^2, 256 -> 256 at 128^2): 256x256 / 2 stages 870-990 TFLOP/s, 128x128 690-820, 256x256 with BK 32 and
    // four stages 780-900; for N = 128 the 256x128 tile loses to 128x128 (660 vs 690)
    if (BK == 64 && tile_dbg != 1 && N > 128 && wgs(256, 256) >= 200) FMI_LAUNCH_B(TB256x256);
    else if (wgs(128, 128) >= 512) FMI_LAUNCH_B(TB128x128);
    else if (wgs(64, 128) >= 256) FMI_LAUNCH_B(TB64x128);
    else FMI_LAUNCH_B(TB64x64);
  }
#undef FMI_LAUNCH_B
  return fmi_launch_status();
}

static int check_desc_b(const fmi_conv_desc* d) {
  if (!d) return FMI_ERR_BAD_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->pad < 0 ||
      d->x_cstride < d->C || d->y_cstride < d->K)
    return FMI_ERR_BAD_ARG;
  if (d->dil > 1) return FMI_ERR_UNSUPPORTED;  // dilated convolutions (modules/drn.py) exist in the fp32 family only
  if (d->OH != (d->H + 2 * d->pad - d->kh) / d->stride + 1 || d->OW != (d->W + 2 * d->pad - d->kw) / d->stride + 1) return FMI_ERR_BAD_ARG;
  if (d->OH <= 0 || d->OW <= 0) return FMI_ERR_BAD_ARG;
  if (d->pad_mode != 0) return FMI_ERR_UNSUPPORTED;
  if ((int64_t)d->N * d->H * d->W > 0x7fffffffLL / 2 || (int64_t)d->N * d->OH * d->OW > 0x7fffffffLL / 2 ||
      (int64_t)d->kh * d->kw * d->C > 0x3fffffffLL || (int64_t)d->kh * d->kw * d->K > 0x3fffffffLL)
    return FMI_ERR_UNSUPPORTED;
  return FMI_OK;
}

extern "C" int fmi_conv2d_bf16_supported(const fmi_conv_desc* d) {
  if (check_desc_b(d) != FMI_OK) return 0;
  return d->C % 32 == 0 && d->K % 32 == 0 && d->x_cstride % 8 == 0 && d->y_cstride % 8 == 0;
}

static int conv2d_fwd_bf16_impl(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* wnk, const float* colscale, uint16_t* y, float* ws,
                               int64_t ws_floats, const EpActB* act, void* stream) {
  int rc = check_desc_b(d);
  if (rc) return rc;
  if (!x || !wnk || !y) return FMI_ERR_BAD_ARG;
  if (d->C % 32 != 0 || d->x_cstride % 8 != 0 || !aligned16b(x) || !aligned16b(wnk)) return FMI_ERR_UNSUPPORTED;
  ConvGeom g{};
  g.N = d->N; g.IH = d->H; g.IW = d->W; g.C = d->C; g.cstride = d->x_cstride;
  g.GH = d->OH; g.GW = d->OW; g.S = d->stride;
  g.nty = d->kh; g.ntx = d->kw; g.dy0 = -d->pad; g.dx0 = -d->pad; g.ystep = 1; g.xstep = 1;
  g.kh0 = 0; g.kw0 = 0; g.khstep = 1; g.kwstep = 1; g.kw = d->kw;
  g.pad_mode = 0; g.vec = d->C % 64 == 0 ? 6 : 5;  // log2 of the reduction tile
  g.dGW = make_fastdiv(g.GW); g.dG = make_fastdiv(g.GH * g.GW); g.dC = make_fastdiv(g.nty * g.ntx); g.dntx = make_fastdiv(g.ntx);
  g.img_bs = 0;
  ConvSetB set{};
  set.ph[0].la = ConvKB{x, g};
  set.ph[0].lb = ConvWKB{wnk, g, d->K, d->kh * d->kw};
  set.ph[0].ep = ConvEpB{y, colscale, d->OH, d->OW, 1, 0, 0, d->OH, d->OW, d->y_cstride, d->K, g.dGW, g.dG,
                         (d->K % 4 == 0 && d->y_cstride % 4 == 0 && ((uintptr_t)y & 7) == 0) ? 1 : 0};
  set.ph[0].M = g.Mdim();
  set.ph[0].K = g.Kdim();
  const int64_t out_elems = (int64_t)d->N * d->OH * d->OW * d->y_cstride;
  if (ws && (((uintptr_t)ws & 15) || ((uintptr_t)y & 7))) ws = nullptr;
  if (d->C % 64 == 0) return launch_conv_bf16<64>(set, 1, d->K, y, ws, ws_floats, out_elems, (hipStream_t)stream, act);
  return launch_conv_bf16<32>(set, 1, d->K, y, ws, ws_floats, out_elems, (hipStream_t)stream, act);
}

extern "C" int fmi_conv2d_fwd_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* wnk, const float* colscale, uint16_t* y,
                                   float* ws, int64_t ws_floats, void* stream) {
  return conv2d_fwd_bf16_impl(d, x, wnk, colscale, y, ws, ws_floats, nullptr, stream);
}
/* y = lrelu(conv(x, W) * colscale[n][k] + nw[0] * noise[n][oy][ox] + bias[k], slope) * gain: a StyledConv without upsampling in one
 * launch (stylegan2/model.py:241-279 ModulatedConv2d on pre-scaled activations, :250-252 demodulation, :282-294 NoiseInjection,
 * op/fused_act.py:30-37 FusedLeakyReLU) -- the output stage of the eight-phase kernel (csrc/conv_bf16_8ph.h).  colscale / noise / bias
 * may be NULL; FMI_ERR_UNSUPPORTED where that kernel does not apply (C % 64, K % 4 and K > 64, 8-byte aligned y, 16-byte colscale / bias). */
extern "C" int fmi_conv2d_fwd_act_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* wnk, const float* colscale, const float* noise,
                                       const float* nw, const float* bias, float slope, float gain, uint16_t* y, void* stream) {
  if (noise && !nw) return FMI_ERR_BAD_ARG;
  if (bias && ((uintptr_t)bias & 15)) return FMI_ERR_UNSUPPORTED;
  EpActB act{noise, nw, bias, slope, gain, 1};
  return conv2d_fwd_bf16_impl(d, x, wnk, colscale, y, nullptr, 0, &act, stream);
}
/* dx = adjoint of the convolution described by d applied to dy; wck = weights packed [C][kh*kw][K] */
extern "C" int fmi_conv2d_dgrad_bf16(const fmi_conv_desc* d, const uint16_t* dy, const uint16_t* wck, const float* colscale, uint16_t* dx,
                                     float* ws, int64_t ws_floats, void* stream) {
  int rc = check_desc_b(d);
  if (rc) return rc;
  if (!dy || !wck || !dx) return FMI_ERR_BAD_ARG;
  if (d->K % 32 != 0 || d->y_cstride % 8 != 0 || !aligned16b(dy) || !aligned16b(wck)) return FMI_ERR_UNSUPPORTED;
  const int s = d->stride;
  if (s > 2) return FMI_ERR_UNSUPPORTED;  // at most four sub-pixel phases per launch
  ConvSetB set{};
  int nph = 0;
  for (int py = 0; py < s; ++py) {
    for (int px = 0; px < s; ++px) {
      const int GH = (d->H - py + s - 1) / s, GW = (d->W - px + s - 1) / s;
      if (GH <= 0 || GW <= 0) continue;
      ConvGeom g{};
      g.N = d->N; g.IH = d->OH; g.IW = d->OW; g.C = d->K; g.cstride = d->y_cstride;
      g.GH = GH; g.GW = GW; g.S = 1;
      g.kh0 = (py + d->pad) % s; g.kw0 = (px + d->pad) % s;
      g.nty = g.kh0 < d->kh ? (d->kh - g.kh0 + s - 1) / s : 0;
      g.ntx = g.kw0 < d->kw ? (d->kw - g.kw0 + s - 1) / s : 0;
      g.dy0 = (py + d->pad - g.kh0) / s; g.dx0 = (px + d->pad - g.kw0) / s;
      g.ystep = -1; g.xstep = -1; g.khstep = s; g.kwstep = s; g.kw = d->kw;
      g.pad_mode = 0; g.vec = d->K % 64 == 0 ? 6 : 5;  // log2 of the reduction tile
      g.dGW = make_fastdiv(GW); g.dG = make_fastdiv(GH * GW);
      g.dntx = make_fastdiv(g.ntx > 0 ? g.ntx : 1);
      g.img_bs = 0;
      if (g.nty == 0 || g.ntx == 0) { g.nty = 0; g.ntx = 1; }
      g.dC = make_fastdiv(g.nty * g.ntx > 0 ? g.nty * g.ntx : 1);
      ConvPhaseB& P = set.ph[nph++];
      P.la = ConvKB{dy, g};
      P.lb = ConvWKB{wck, g, d->C, d->kh * d->kw};
      P.ep = ConvEpB{dx, colscale, GH, GW, s, py, px, d->H, d->W, d->x_cstride, d->C, g.dGW, g.dG,
                     (d->C % 4 == 0 && d->x_cstride % 4 == 0 && ((uintptr_t)dx & 7) == 0) ? 1 : 0};
      P.M = g.Mdim();
      P.K = g.Kdim();
    }
  }
  if (nph == 0) return FMI_OK;
  const int64_t out_elems = (int64_t)d->N * d->H * d->W * d->x_cstride;
  if (ws && (((uintptr_t)ws & 15) || ((uintptr_t)dx & 7))) ws = nullptr;
  return d->K % 64 == 0 ? launch_conv_bf16<64>(set, nph, d->C, dx, ws, ws_floats, out_elems, (hipStream_t)stream)
                        : launch_conv_bf16<32>(set, nph, d->C, dx, ws, ws_floats, out_elems, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight packs: src fp32 [T][A][B] (the fp32 packs wf[tap][C][K] / wt[tap][K][C]) -> bf16 [B][T][A]
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_bta_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int T, int A, int B,
                                                            int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int a = (int)(i % A);
    const int64_t r = i / A;
    const int t = (int)(r % T), b = (int)(r / T);
    dst[i] = f2bf(src[((int64_t)t * A + a) * B + b]);
  }
}
extern "C" int fmi_pack_weight_bf16(const float* src_tab, uint16_t* dst_bta, int T, int A, int B, void* stream) {
  if (!src_tab || !dst_bta || T <= 0 || A <= 0 || B <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)T * A * B;
  hipLaunchKernelGGL(pack_bta_bf16_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src_tab, dst_bta, T, A, B, total);
  return fmi_launch_status();
}
// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient: dwf[tap][c][k] += sum over pixels x[pixel + tap][c] * dy[pixel][k]   (fp32 atomics across the pixel splits)
// ---------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct WgArgsB {
  const bf16_t* x;
  const bf16_t* dy;
  float* dwf;
  ConvGeom g;  // forward geometry: anchors = output pixels, all kh*kw taps
  int Kout, ycs, Mrows, P, kchunk;
  int tiles, ksplit, xcd_splits;
};

// LDS image of one operand tile: [32-row group][64 pixels][32 rows] bf16 -- a wave instruction of the copy fills 16 pixels x 64
// bytes of one group, so a thread's copies share ONE pixel decode; a transposed read of four pixel rows is 256 contiguous bytes.
template <class T>
__global__ void __launch_bounds__(T::NT) wgrad_bf16_kernel(WgArgsB a, int tiles_n) {
  const float* const zchunk = fmi_zero_chunk_ptr();  // the zero chunk's address: read from the GOT ONCE (see fmi_zero_chunk_ptr)
  constexpr int BM = T::BM, BN = T::BN, BK = 64;
  constexpr int WPP = T::NT / 256;  // waves per 16-pixel block of the tile (8-wave tiles: two, each filling every other row group)
  constexpr int NGA = BM / 32 / WPP, NGB = BN / 32 / WPP;
  static_assert(NGA >= 1 && NGB >= 1, "every wave copies at least one group of each operand");
  constexpr int STAGE_B = (BM + BN) * BK * 2;  // bytes
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // Workgroup -> (pixel split, tile): the workgroups of ONE split share their pixel range (every tile re-reads it for its own rows /
  // columns), so a split is pinned to one XCD (linear id % 8 = XCD): its tiles stream through the range together and the XCD's L2
  // serves all re-reads.  With tiles fastest across ALL XCDs, FETCH_SIZE was 2 GB per launch for 134 MB of operands (5 TB/s of fabric
  // traffic at 0.4 ms).
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

  const int kr = 16 * (wid & 3) + (lane >> 2), cq = lane & 3, gsel = wid >> 2;
  // A copies: group j = rows m0 + 32 j .. +31 lie inside one tap (C % 32 == 0): wave-uniform tap, per-thread channel
  int a_dy[NGA], a_dx[NGA];
  int64_t a_off[NGA];
#pragma unroll
  for (int j = 0; j < NGA; ++j) {
    const int row = m0 + 32 * (j * WPP + gsel);
    const int t = (int)fdiv((uint32_t)row, g.dC);
    const int i = (int)fdiv((uint32_t)t, g.dntx), jx = t - i * g.ntx;
    a_dy[j] = row < a.Mrows ? g.dy0 + i : -0x20000000;
    a_dx[j] = g.dx0 + jx;
    a_off[j] = ((int64_t)a_dy[j] * g.IW + a_dx[j]) * g.cstride + (row - t * g.C) + 8 * cq;
  }
  int b_col[NGB];
#pragma unroll
  for (int j = 0; j < NGB; ++j) {
    const int col = n0 + 32 * (j * WPP + gsel) + 8 * cq;
    b_col[j] = col < a.Kout ? col : -1;
  }

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  auto glds16 = [&](const void* gp, uint32_t dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gp), "s"(dst)
                 : "memory");
  };
  auto issue = [&](int k0, int st) {
    const uint32_t sa = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)st * STAGE_B + (uint32_t)(wid & 3) * 1024u + (uint32_t)gsel * 4096u);
    const uint32_t sb = sa + BM * BK * 2;
    const int pix = k0 + kr;
    const bool pv = pix < k_end;
    const uint32_t n = fdiv((uint32_t)pix, g.dG);
    const uint32_t rem = (uint32_t)pix - n * (uint32_t)(g.GH * g.GW);
    const uint32_t gy = fdiv(rem, g.dGW);
    const uint32_t gx = rem - gy * (uint32_t)g.GW;
    const int iy0 = (int)gy * g.S, ix0 = (int)gx * g.S;
    const int64_t xb = ((int64_t)((int)n * g.IH + iy0) * g.IW + ix0) * g.cstride;
#pragma unroll
    for (int j = 0; j < NGA; ++j) {
      const bool ok = pv && (unsigned)(iy0 + a_dy[j]) < (unsigned)g.IH && (unsigned)(ix0 + a_dx[j]) < (unsigned)g.IW;
      const void* gp = ok ? (const void*)(a.x + xb + a_off[j]) : (const void*)zchunk;
      glds16(gp, sa + j * WPP * 4096);
    }
#pragma unroll
    for (int j = 0; j < NGB; ++j) {
      const bool ok = pv && b_col[j] >= 0;
      const void* gp = ok ? (const void*)(a.dy + (int64_t)pix * a.ycs + b_col[j]) : (const void*)zchunk;
      glds16(gp, sb + j * WPP * 4096);
    }
  };
  // transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, rows 4p..4p+3 of the group's 16
  const int i16 = lane & 15;
  const uint32_t lbase = (uint32_t)((8 * lh + (i16 >> 2)) * 64 + 32 * ((lane >> 4) & 1) + 8 * (i16 & 3));
  auto frag = [&](uint32_t img, int grp, int s) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const uint32_t ad = img + (uint32_t)grp * 4096u + lbase + (uint32_t)s * 1024u;
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
  auto compute = [&](int st) {
    const uint32_t sa = lds0 + (uint32_t)st * STAGE_B, sb = sa + BM * BK * 2;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 fa[T::TM], fb[T::TN];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) fa[i] = frag(sa, wm / 32 + i, s);
#pragma unroll
      for (int j = 0; j < T::TN; ++j) fb[j] = frag(sb, wn / 32 + j, s);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = (k_end - k_begin + BK - 1) / BK;
  if (nt > 0) issue(k_begin, 0);
  int st = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < nt) issue(k_begin + (t + 1) * BK, st ^ 1);
    compute(st);
    st ^= 1;
  }
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

#include "wgrad_bf16_8ph.h"

/* dwf[tap][C][K] (fp32, the layout of fmi_conv2d_wgrad_f32) += x^T dy; caller zeroes dwf */
extern "C" int fmi_conv2d_wgrad_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dwf, void* stream) {
  int rc = check_desc_b(d);
  if (rc) return rc;
  if (!x || !dy || !dwf) return FMI_ERR_BAD_ARG;
  if (d->C % 32 != 0 || d->K % 8 != 0 || d->x_cstride % 8 != 0 || d->y_cstride % 8 != 0 || !aligned16b(x) || !aligned16b(dy))
    return FMI_ERR_UNSUPPORTED;
  WgArgsB a{};
  a.x = x; a.dy = dy; a.dwf = dwf;
  ConvGeom& g = a.g;
  g.N = d->N; g.IH = d->H; g.IW = d->W; g.C = d->C; g.cstride = d->x_cstride;
  g.GH = d->OH; g.GW = d->OW; g.S = d->stride;
  g.nty = d->kh; g.ntx = d->kw; g.dy0 = -d->pad; g.dx0 = -d->pad; g.ystep = 1; g.xstep = 1;
  g.kh0 = 0; g.kw0 = 0; g.khstep = 1; g.kwstep = 1; g.kw = d->kw;
  g.dGW = make_fastdiv(g.GW); g.dG = make_fastdiv(g.GH * g.GW); g.dC = make_fastdiv(g.C); g.dntx = make_fastdiv(g.ntx);
  a.Kout = d->K; a.ycs = d->y_cstride; a.Mrows = d->kh * d->kw * d->C; a.P = d->N * d->OH * d->OW;
  // the eight-phase kernel (wgrad_bf16_8ph.h): 256 x 256 tiles, one workgroup per CU -- for wide layers with a long pixel range
  {
    const int mode = bf16_tile_mode();
    const int64_t tm8 = ceil_div64(a.Mrows, 256), tn8 = ceil_div64(d->K, 256);
    const bool fits = d->K >= 256 && tn8 * 256 - d->K <= d->K / 8 && tm8 * 256 - a.Mrows <= a.Mrows / 8 && tm8 * tn8 <= 256 && a.P >= 4096 &&
                      (int64_t)d->kh * d->kw * d->C < (1ll << 30) && (int64_t)d->N * d->H * d->W * d->x_cstride < (1ll << 31);
    if (fits && !fmi_det() && mode != 1 && mode != 2) {
      Wg8Args w{};
      w.x = x; w.dy = dy; w.dwf = dwf; w.g = g;
      w.Kout = d->K; w.ycs = d->y_cstride; w.Mrows = a.Mrows; w.P = a.P;
      w.tiles = (int)(tm8 * tn8); w.tiles_n = (int)tn8;
      int64_t ks = 256 / w.tiles;
      if (ks > a.P / 1024) ks = a.P / 1024;
      if (ks < 1) ks = 1;
      w.kchunk = (int)(ceil_div64(ceil_div64(a.P, ks), 64) * 64);
      w.ksplit = (int)ceil_div64(a.P, w.kchunk);
      hipLaunchKernelGGL(wgrad_bf16_8ph_kernel, dim3((unsigned)(w.tiles * w.ksplit)), dim3(512), 0, (hipStream_t)stream, w);
      return fmi_launch_status();
    }
  }
  // the 8-wave 256x256 tile (the kernel takes it: WPP = 2) measured SLOWER here than 128x128 (574 vs 770 TFLOP/s at 512 -> 512, 64^2):
  // one workgroup per CU and a 16-accumulator atomic epilogue per split leave the CU idle between tiles
  const int bn = d->K <= 32 ? 32 : (d->K <= 64 ? 64 : 128);
  const int64_t tm = ceil_div64(a.Mrows, 128), tn = ceil_div64(d->K, bn);
  int64_t ksplit = 2048 / (tm * tn);
  const int64_t kmax = a.P / 1024;
  if (ksplit > kmax) ksplit = kmax;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 65535) ksplit = 65535;
  if (ksplit >= 6) ksplit = (ksplit + 4) / 8 * 8;  // splits are dealt to the 8 XCDs in groups of 8: keep the groups full
  if (fmi_det()) ksplit = 1;                       // reproducible mode
  a.kchunk = (int)(ceil_div64(ceil_div64(a.P, ksplit), 64) * 64);
  ksplit = ceil_div64(a.P, a.kchunk);
  a.tiles = (int)(tm * tn);
  a.ksplit = (int)ksplit;
  static const bool xcd_off = getenv("FMI_WGRAD_XCD_OFF") != nullptr;  // debug: splits in grid.y, tiles over all XCDs
  a.xcd_splits = (ksplit >= 8 && !xcd_off) ? 1 : 0;
  const int64_t nwg = a.xcd_splits ? tm * tn * ceil_div64(ksplit, 8) * 8 : tm * tn;
  if (nwg > 0x7fffffffLL) return FMI_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)nwg, a.xcd_splits ? 1u : (unsigned)ksplit);
  hipStream_t st = (hipStream_t)stream;
  if (bn == 32) hipLaunchKernelGGL((wgrad_bf16_kernel<TB128x32>), grid, dim3(256), 0, st, a, (int)tn);
  else if (bn == 64) hipLaunchKernelGGL((wgrad_bf16_kernel<TB128x64>), grid, dim3(256), 0, st, a, (int)tn);
  else hipLaunchKernelGGL((wgrad_bf16_kernel<TB128x128>), grid, dim3(256), 0, st, a, (int)tn);
  return fmi_launch_status();
}
#endif  // FMI_HOST_EMU
