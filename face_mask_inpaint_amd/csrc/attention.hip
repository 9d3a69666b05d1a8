// Fused self-attention forward for the PICNet attention blocks (example_guided_att.py:21-33, base_function.py:420-437):
//      A = softmax_j(q_i . q_j)   (queries = keys, no scaling)      O = A V        V = [V1 | V2] along channels
// in exact fp32 on the matrix cores, flash style: the [T x T] map (1 GiB per image at T = 16384) is never formed.
//
// MI355X mapping
//  * one 256-thread workgroup = 4 waves = 128 queries; each wave owns 32 queries, ONE PER LANE COLUMN: the score tile is
//    computed transposed, S^T[key][query] = K Q^T (A = K tile from LDS, B = the wave's Q fragment held in registers for
//    the whole kernel), so a lane holds 16 keys of its own query.  Row max / row sum are then in-lane reductions plus one
//    exchange with lane^32, and the exponentiated tile is ALREADY the B operand of the second product
//    O^T[c][query] += V^T[c][key] P^T[key][query] (k-index permuted identically on both operands): P never touches LDS.
//  * O^T accumulators: C/32 tiles x 16 registers per lane (128 for 256 channels); one wave per SIMD, so the kernel runs
//    with the 512-register budget (launch_bounds(256, 1)).
//  * K is staged transposed ([d][key], +1 pad) and V row-major in LDS, double buffered, filled through registers
//    (prefetch of tile t+1 is issued before the 160 MFMAs of tile t), one barrier per 32-key tile.
//  * online softmax with a LAZY reference maximum: the accumulators are rescaled only when a tile's maximum exceeds the
//    reference by more than 20 (fp32 has the range; p <= e^20), which happens in the first tiles only; the branch is
//    wave-uniform.
#include "common.h"
#include "x6.h"
#include <cstdlib>

// > 64 KB of dynamic LDS must be opted into per kernel AND per device (hipFuncSetAttribute acts on the current device's code object):
// one atomic flag per (kernel instantiation, device ordinal), safe from the forward and the autograd thread alike
struct fmi_attr_flags {
  unsigned char done[64];
};
static inline bool fmi_attr_needed(fmi_attr_flags& f, int& dev) {
  dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  return !__atomic_load_n(&f.done[dev], __ATOMIC_ACQUIRE);
}
static inline void fmi_attr_mark(fmi_attr_flags& f, int dev) {
  if (dev >= 0 && dev < 64) __atomic_store_n(&f.done[dev], (unsigned char)1, __ATOMIC_RELEASE);
}

#define ATT_THR 20.0f


template <int D, int CT, int NKL, int NVL, int NTH>
__device__ __forceinline__ void att_gload(float4 (&rk)[NKL], float4 (&rv)[NVL], const float* __restrict__ qb,
                                          const float* __restrict__ v1b, const float* __restrict__ v2b, int C1, int C2, int k0,
                                          int tid) {
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int f = tid + NTH * i;
    const int key = f / (D / 4), dq = f % (D / 4);
    rk[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(k0 + key) * D + dq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < NVL; ++i) {
    const int f = tid + NTH * i;
    const int key = f / (CT / 4), c = (f % (CT / 4)) * 4;
    rv[i] = (c < C1) ? *reinterpret_cast<const float4*>(v1b + (int64_t)(k0 + key) * C1 + c)
                     : *reinterpret_cast<const float4*>(v2b + (int64_t)(k0 + key) * C2 + (c - C1));
  }
}
template <int D, int CT, int NKL, int NVL, int LDK, int NTH>
__device__ __forceinline__ void att_lstore(const float4 (&rk)[NKL], const float4 (&rv)[NVL], float* __restrict__ kt,
                                           float* __restrict__ vs, int tid) {
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int f = tid + NTH * i;
    if (f < 8 * D) {
      const int key = f / (D / 4), dd = (f % (D / 4)) * 4;
      kt[(dd + 0) * LDK + key] = rk[i].x;
      kt[(dd + 1) * LDK + key] = rk[i].y;
      kt[(dd + 2) * LDK + key] = rk[i].z;
      kt[(dd + 3) * LDK + key] = rk[i].w;
    }
  }
#pragma unroll
  for (int i = 0; i < NVL; ++i) {
    const int f = tid + NTH * i;
    *reinterpret_cast<float4*>(vs + f * 4) = rv[i];  // f*4 = key*CT + c
  }
}

// NW waves per workgroup, each owning 32 queries; all share the staged key / value tiles.  NW = 8 for long sequences: the 82 KB of LDS
// allow one workgroup per CU, and with four waves every SIMD held ONE wave whose softmax (exp, max, rescale) left the matrix pipe
// idle; eight waves put two on each SIMD.  NW = 2 for short ones (T = 1024: 64 workgroups of four waves filled a quarter of the chip).
template <int D, int NCT, int NW>
__global__ void __launch_bounds__(NW * 64, 1) attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                          const float* __restrict__ v2, float* __restrict__ o1,
                                                          float* __restrict__ o2, float* __restrict__ lse, int T, int C1,
                                                          int C2) {
  constexpr int CT = NCT * 32;        // total value channels
  constexpr int LDK = 33;             // Kt row pitch (floats)
  constexpr int KT_FLOATS = D * LDK, VS_FLOATS = 32 * CT;
  constexpr int NTH = NW * 64;
  constexpr int NKL = (8 * D + NTH - 1) / NTH;                 // float4 loads per thread for a K tile
  constexpr int NVL = (8 * CT) / NTH;                          // float4 loads per thread for a V tile
  static_assert((8 * CT) % NTH == 0, "whole float4 passes over the V tile");
  __shared__ __attribute__((aligned(16))) float lds[2 * (KT_FLOATS + VS_FLOATS)];
  float* Kt = lds;
  float* Vs = lds + 2 * KT_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y;
  const int q0 = blockIdx.x * (NW * 32) + wid * 32;
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;

  // this lane's query fragment: B[k = dd][j = query], dd = 2s + lh
  float qf[D / 2];
#pragma unroll
  for (int s = 0; s < D / 2; ++s) qf[s] = qb[(int64_t)(q0 + l31) * D + 2 * s + lh];

  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float mref = -INFINITY, lsum = 0.f;

  float4 rk[NKL], rv[NVL];
  att_gload<D, CT, NKL, NVL, NTH>(rk, rv, qb, v1b, v2b, C1, C2, 0, tid);
  att_lstore<D, CT, NKL, NVL, LDK, NTH>(rk, rv, Kt, Vs, tid);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32) {
    // unconditional prefetch (the last iteration re-reads its own tile): a conditional one sends rk/rv to scratch
    att_gload<D, CT, NKL, NVL, NTH>(rk, rv, qb, v1b, v2b, C1, C2, k0 + 32 < T ? k0 + 32 : k0, tid);
    const float* kt = Kt + buf * KT_FLOATS + l31;
    const float* vs = Vs + buf * VS_FLOATS + l31;
    // S^T[key][query] for 32 keys x this wave's 32 queries
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[(2 * s + lh) * LDK], qf[s], st, 0, 0, 0);
    // online softmax, one query per lane column (keys split over the lane pair l, l^32)
    float tmax = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__any(tmax > mref + ATT_THR)) {
      const float mnew = fmaxf(mref, tmax);
      const float alpha = __expf(mref - mnew);
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] *= alpha;
      lsum *= alpha;
      mref = mnew;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __expf(st[r] - mref);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    lsum += psum;
    // O^T[c][query] += V^T[c][key] P^T[key][query]; MFMA step s of half lh carries key (s&3) + 8*(s>>2) + 4*lh
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = (s & 3) + 8 * (s >> 2) + 4 * lh;
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * CT + c * 32], st[s], acc[c], 0, 0, 0);
    }
    att_lstore<D, CT, NKL, NVL, LDK, NTH>(rk, rv, Kt + (buf ^ 1) * KT_FLOATS, Vs + (buf ^ 1) * VS_FLOATS, tid);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: register r of tile c is channel c*32 + (r&3) + 8*(r>>2) + 4*lh of query q0 + l31
  const float inv = 1.f / lsum;
  const int64_t row = (int64_t)n * T + q0 + l31;
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int cbase = c * 32;
    float* ob = (cbase < C1) ? o1 + row * C1 + cbase : o2 + row * C2 + (cbase - C1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o;
      o.x = acc[c][4 * g + 0] * inv;
      o.y = acc[c][4 * g + 1] * inv;
      o.z = acc[c][4 * g + 2] * inv;
      o.w = acc[c][4 * g + 3] * inv;
      *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) = o;
    }
  }
  if (lh == 0 && lse) lse[row] = mref + logf(lsum);
}


// =====================================================================================================
// Forward on the bf16 matrix pipe (x6.h): the same flash structure, every product as six bf16 MFMAs of K = 16 on exact three-way
// splits of the fp32 operands.
//  * the K / V tile of 32 keys is split ONCE, on its way from the staging registers into LDS, into three bf16 images each:
//      K pieces  [32 keys][D bf16],  row pitch 2 D + 16 bytes: the A fragment (key = lane, 8 consecutive d) is one ds_read_b128;
//      V pieces  [32 keys][CT bf16], row pitch 2 CT + 64 bytes: the A fragment of O^T += V^T P^T (channel = lane, keys in the
//                reduction) is two ds_read_b64_tr_b16 -- the hardware transposes a 4-key x 16-channel block per 16 lanes;
//  * Q pieces stay in registers (B operand of S^T = K Q^T), and the exponentiated tile, still one query per lane column, is cut
//    into its three pieces in registers: element j of lane half h of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3), the order
//    the score tile's registers already have, so the V reads simply address those keys;
//  * per 32-key tile and wave: 6 x (D / 16 + 2 NCT) MFMAs (120 at D = 64, C = 256: 3 840 matrix-pipe cycles against 10 240 for the
//    160 fp32 MFMAs), ~110 VALU instructions for the K / V split and 88 for P.
// =====================================================================================================
template <int D, int NCT, int NW>
__global__ void __launch_bounds__(NW * 64, 1) attn_fwd_x6_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                                 const float* __restrict__ v2, float* __restrict__ o1,
                                                                 float* __restrict__ o2, float* __restrict__ lse, int T, int C1,
                                                                 int C2) {
  constexpr int CT = NCT * 32;
  constexpr int KP = 2 * D + 16, VP = 2 * CT + 64;          // row pitches (bytes)
  constexpr int KIMG = 32 * KP, VIMG = 32 * VP;             // one piece image
  constexpr int STAGE = 3 * KIMG + 3 * VIMG;                // bytes per stage
  constexpr int NTH = NW * 64;
  constexpr int NKL = (8 * D + NTH - 1) / NTH;              // float4 loads per thread for a K tile
  constexpr int NVL = (8 * CT) / NTH;                       // float4 loads per thread for a V tile
  static_assert((8 * CT) % NTH == 0, "whole float4 passes over the V tile");
  static_assert(D % 16 == 0, "whole bf16 k-steps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x6[];
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem_x6;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y;
  const int q0 = blockIdx.x * (NW * 32) + wid * 32;
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;

  // this lane's query pieces: B[k = d][col = query], d = 16 kk + 8 lh + j
  bf16x8_t qp[D / 16][3];
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk) {
    const float4 a = *reinterpret_cast<const float4*>(qb + (int64_t)(q0 + l31) * D + 16 * kk + 8 * lh);
    const float4 b = *reinterpret_cast<const float4*>(qb + (int64_t)(q0 + l31) * D + 16 * kk + 8 * lh + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    split3_bf16(f, qp[kk]);
  }

  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float mref = -INFINITY, lsum = 0.f;

  // staging: split each float4 into three 8-byte pieces and store them into the images of stage `st`
  float4 rk[NKL], rv[NVL];
  auto lstore = [&](int st) {
    unsigned char* base = smem_x6 + st * STAGE;
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int f = tid + NTH * i;
      if (f < 8 * D) {
        const int key = f / (D / 4), dq = f % (D / 4);
        uint32_t a0, a1, a2, b0, b1, b2;
        split3_pair(rk[i].x, rk[i].y, a0, a1, a2);
        split3_pair(rk[i].z, rk[i].w, b0, b1, b2);
        unsigned char* d = base + key * KP + dq * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(d + KIMG) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(d + 2 * KIMG) = make_uint2(a2, b2);
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + NTH * i;
      const int key = f / (CT / 4), c4 = f % (CT / 4);
      uint32_t a0, a1, a2, b0, b1, b2;
      split3_pair(rv[i].x, rv[i].y, a0, a1, a2);
      split3_pair(rv[i].z, rv[i].w, b0, b1, b2);
      unsigned char* d = base + 3 * KIMG + key * VP + c4 * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + VIMG) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 2 * VIMG) = make_uint2(a2, b2);
    }
  };
  // transposed V fragment reads: lane 4 q + p of a 16-lane group addresses key row q, channels 4 p .. 4 p + 3 of the group's 16
  const int i16 = lane & 15;
  const uint32_t vlane = (uint32_t)((4 * lh + (i16 >> 2)) * VP + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2);
  const uint32_t klane = (uint32_t)(l31 * KP + 16 * lh);

  att_gload<D, CT, NKL, NVL, NTH>(rk, rv, qb, v1b, v2b, C1, C2, 0, tid);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32) {
    // unconditional prefetch (the last iteration re-reads its own tile): a conditional one sends rk/rv to scratch
    att_gload<D, CT, NKL, NVL, NTH>(rk, rv, qb, v1b, v2b, C1, C2, k0 + 32 < T ? k0 + 32 : k0, tid);
    const uint32_t kimg = lds0 + (uint32_t)(buf * STAGE) + klane;
    const uint32_t vimg = lds0 + (uint32_t)(buf * STAGE + 3 * KIMG) + vlane;
    // S^T[key][query] for 32 keys x this wave's 32 queries
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
      bf16x8_t kp[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        typedef __attribute__((address_space(3))) bf16x8_t* lp8;
        kp[pc] = *(lp8)(uintptr_t)(kimg + (uint32_t)(pc * KIMG + kk * 32));
      }
      st = mfma_x6(kp, qp[kk], st);
    }
    // online softmax, one query per lane column (keys split over the lane pair l, l^32)
    float tmax = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__any(tmax > mref + ATT_THR)) {
      const float mnew = fmaxf(mref, tmax);
      const float alpha = __expf(mref - mnew);
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] *= alpha;
      lsum *= alpha;
      mref = mnew;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __expf(st[r] - mref);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    lsum += psum;
    // O^T[c][query] += V^T[c][key] P^T[key][query]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t pp[3];
      {
        const float f[8] = {st[8 * s], st[8 * s + 1], st[8 * s + 2], st[8 * s + 3], st[8 * s + 4], st[8 * s + 5], st[8 * s + 6], st[8 * s + 7]};
        split3_bf16(f, pp);
      }
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        bf16x8_t vp[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          typedef short s16x4_t __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4_t* lp4;
          const uint32_t ad = vimg + (uint32_t)(pc * VIMG + 16 * s * VP + 64 * c);
          union {
            s16x4_t h[2];
            bf16x8_t v;
          } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)ad);
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)(ad + 8 * VP));
          vp[pc] = u.v;
        }
        acc[c] = mfma_x6(vp, pp, acc[c]);
      }
    }
    lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: register r of tile c is channel c*32 + (r&3) + 8*(r>>2) + 4*lh of query q0 + l31
  const float inv = 1.f / lsum;
  const int64_t row = (int64_t)n * T + q0 + l31;
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int cbase = c * 32;
    float* ob = (cbase < C1) ? o1 + row * C1 + cbase : o2 + row * C2 + (cbase - C1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o;
      o.x = acc[c][4 * g + 0] * inv;
      o.y = acc[c][4 * g + 1] * inv;
      o.z = acc[c][4 * g + 2] * inv;
      o.w = acc[c][4 * g + 3] * inv;
      *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) = o;
    }
  }
  if (lh == 0 && lse) lse[row] = mref + logf(lsum);
}

extern "C" int fmi_attention_fwd_f32(const float* q, const float* v1, const float* v2, float* o1, float* o2, float* lse,
                                     int N, int T, int D, int C1, int C2, void* stream) {
  if (!q || !v1 || !o1 || N <= 0 || T <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && (!v2 || !o2))) return FMI_ERR_BAD_ARG;
  if (T % 128 != 0 || C1 % 32 != 0 || C2 % 32 != 0 || N > 65535) return FMI_ERR_UNSUPPORTED;
  static const int nw_dbg = getenv("FMI_ATT_NW") ? atoi(getenv("FMI_ATT_NW")) : 0;  // debug: force 2 / 4 / 8 waves per workgroup
  int nw = (T % 256 == 0 && (int64_t)(T / 256) * N >= 256) ? 8 : ((int64_t)(T / 128) * N < 256 ? 2 : 4);
  if (nw_dbg == 2 || nw_dbg == 4 || (nw_dbg == 8 && T % 256 == 0)) nw = nw_dbg;
  if ((((uintptr_t)q | (uintptr_t)v1 | (uintptr_t)v2 | (uintptr_t)o1 | (uintptr_t)o2) & 15) != 0) return FMI_ERR_BAD_ARG;
  const int nct = (C1 + C2) / 32;
  const dim3 grid(T / (nw * 32), N), block(nw * 64);
  hipStream_t st = (hipStream_t)stream;
#if FMI_X6
#define ATT_LAUNCH1(DD, NN, WW)                                                                                                \
  do {                                                                                                                        \
    constexpr int lds_bytes = 2 * 3 * 32 * ((2 * DD + 16) + (2 * NN * 32 + 64));                                               \
    static fmi_attr_flags attr_set{}; int attr_set_dev;\
    if (fmi_attr_needed(attr_set, attr_set_dev)) {                                                                                                          \
      if (hipFuncSetAttribute((const void*)attn_fwd_x6_kernel<DD, NN, WW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) \
        return FMI_ERR_LAUNCH;                                                                                                \
      fmi_attr_mark(attr_set, attr_set_dev);                                                                                                        \
    }                                                                                                                         \
    hipLaunchKernelGGL((attn_fwd_x6_kernel<DD, NN, WW>), grid, block, lds_bytes, st, q, v1, v2, o1, o2, lse, T, C1, C2);       \
  } while (0)
#else
#define ATT_LAUNCH1(DD, NN, WW) hipLaunchKernelGGL((attn_fwd_kernel<DD, NN, WW>), grid, block, 0, st, q, v1, v2, o1, o2, lse, T, C1, C2)
#endif
#define ATT_LAUNCH(DD, NN)                 \
  do {                                     \
    if (nw == 8) ATT_LAUNCH1(DD, NN, 8);   \
    else if (nw == 2) ATT_LAUNCH1(DD, NN, 2); \
    else ATT_LAUNCH1(DD, NN, 4);           \
  } while (0)
  if (D == 64 && nct == 8) ATT_LAUNCH(64, 8);
  else if (D == 32 && nct == 8) ATT_LAUNCH(32, 8);
  else if (D == 32 && nct == 4) ATT_LAUNCH(32, 4);
  else if (D == 64 && nct == 4) ATT_LAUNCH(64, 4);
  else if (D == 16 && nct == 2) ATT_LAUNCH(16, 2);
  else if (D == 16 && nct == 4) ATT_LAUNCH(16, 4);
  else return FMI_ERR_UNSUPPORTED;
#undef ATT_LAUNCH
#undef ATT_LAUNCH1
  return fmi_launch_status();
}

// =====================================================================================================
// Backward.  With P = softmax(q q^T), O_i = P V_i, upstream gO_i:
//   dV = P^T gO      dP = gO V^T      dS = P o (dP - delta),  delta_q = sum_c gO[q][c] O[q][c]
//   dQ[q] += sum_key dS[q][key] q[key]   (query side)       dQ[key] += sum_q dS[q][key] q[q]   (key side; K = Q)
// One workgroup owns a block of 32 KEYS (dV and the key-side dQ of those keys stay in registers for the whole kernel)
// and streams all query tiles; P is recomputed from the saved log-sum-exp.  The four waves of the workgroup split the
// REDUCTION dimensions of the two tile products that feed the softmax backward -- S over d, dP over the value
// channels -- and exchange their 32x32 partial tiles through LDS, so no product is computed twice:
//   per (32 queries x 32 keys) pair and wave:  S part 8 + dP part 32 + dV (own 64 channels) 32 + dK or dQ tile 16 MFMAs.
// Scores are computed with the key on the lane, so P and dS are directly the B operands of the dV / dK products; only
// the query-side product needs dS transposed, through a private 4 KB LDS tile.  The query-side dQ tiles are added with
// fp32 atomics (whole 128-byte rows per wave instruction).
// =====================================================================================================
template <int D, int NCT>  // NCT = (C1 + C2) / 32, multiple of 4
__global__ void __launch_bounds__(256, 1) attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                          const float* __restrict__ v2, const float* __restrict__ g1,
                                                          const float* __restrict__ g2, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, float* __restrict__ gv1,
                                                          float* __restrict__ gv2, float* __restrict__ gq, int T, int C1, int C2, int kb0) {
  constexpr int CT = NCT * 32, CW = CT / 4, NCW = CW / 32, DW = D / 4;
  constexpr int LDV = CT + 1, LDQ = D + 1;
  constexpr int NQL = (8 * D) / 256 > 0 ? (8 * D) / 256 : 1, NVL = (8 * CT) / 256;
  constexpr int NDT = D / 32;  // 32-wide d tiles (1 or 2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vj = smem;                   // [32][LDV]
  float* Kj = Vj + 32 * LDV;          // [32][LDQ]
  float* dOi = Kj + 32 * LDQ;         // [32][LDV]
  float* Qi = dOi + 32 * LDV;         // [32][LDQ]
  float* lse_i = Qi + 32 * LDQ;       // [32]
  float* del_i = lse_i + 32;          // [32]
  float* EX = del_i + 32;             // [4][2][16][64]
  float* dsT = EX + 4 * 2 * 1024;     // [2][32][33]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y, j0 = ((int)blockIdx.x + kb0) * 32;  // kb0: key-block offset of this launch (reproducible mode: one key block per launch)
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;
  const float* g1b = g1 + (int64_t)n * T * C1;
  const float* g2b = g2 ? g2 + (int64_t)n * T * C2 : nullptr;
  const float* lseb = lse + (int64_t)n * T;
  const float* delb = delta + (int64_t)n * T;

  // resident key block: V_j, K_j (row-major, odd pitch: conflict-free both along a row and down a column)
  for (int f = tid; f < 8 * CT; f += 256) {
    const int key = f / (CT / 4), c = (f % (CT / 4)) * 4;
    const float4 v = (c < C1) ? *reinterpret_cast<const float4*>(v1b + (int64_t)(j0 + key) * C1 + c)
                              : *reinterpret_cast<const float4*>(v2b + (int64_t)(j0 + key) * C2 + (c - C1));
    float* d = Vj + key * LDV + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  for (int f = tid; f < 8 * D; f += 256) {
    const int key = f / (D / 4), dd = (f % (D / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(qb + (int64_t)(j0 + key) * D + dd);
    float* d = Kj + key * LDQ + dd;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }

  f32x16 acc_dv[NCW], acc_dk;
#pragma unroll
  for (int c = 0; c < NCW; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dv[c][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_dk[r] = 0.f;
  const int cbase = wid * CW;            // this wave's value-channel slice
  const int dbase = wid * DW;            // this wave's d slice of the score product
  const bool dk_role = wid < NDT;        // waves 0..NDT-1: key-side tile wid;  waves 2..2+NDT-1: query-side tile wid-2
  const bool dq_role = wid >= 2 && wid - 2 < NDT;
  const int dt = dk_role ? wid : wid - 2;

  float4 rq[NQL], rg[NVL];
  float rl = 0.f, rd = 0.f;
  auto gload = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (D / 4), dd = (f % (D / 4)) * 4;
      rq[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(i0 + row) * D + dd) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (CT / 4), c = (f % (CT / 4)) * 4;
      rg[i] = (c < C1) ? *reinterpret_cast<const float4*>(g1b + (int64_t)(i0 + row) * C1 + c)
                       : *reinterpret_cast<const float4*>(g2b + (int64_t)(i0 + row) * C2 + (c - C1));
    }
    rl = lseb[i0 + (tid & 31)];
    rd = delb[i0 + (tid & 31)];
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
      if (f < 8 * D) {
        float* d = Qi + (f / (D / 4)) * LDQ + (f % (D / 4)) * 4;
        d[0] = rq[i].x; d[1] = rq[i].y; d[2] = rq[i].z; d[3] = rq[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 256 * i;
      float* d = dOi + (f / (CT / 4)) * LDV + (f % (CT / 4)) * 4;
      d[0] = rg[i].x; d[1] = rg[i].y; d[2] = rg[i].z; d[3] = rg[i].w;
    }
    if (tid < 32) {
      lse_i[tid] = rl;
      del_i[tid] = rd;
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int i0 = 0; i0 < T; i0 += 32) {
    gload(i0 + 32 < T ? i0 + 32 : i0);  // unconditional prefetch of the next query tile into registers
    // ---- 1. partial score / dP tiles over this wave's slice of the reduction dimension; lane = key, registers = queries
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sp[r] = 0.f;
      dp[r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < DW / 2; ++s)
      sp = __builtin_amdgcn_mfma_f32_32x32x2f32(Qi[l31 * LDQ + dbase + 2 * s + lh], Kj[l31 * LDQ + dbase + 2 * s + lh], sp, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < CW / 2; ++s)
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dOi[l31 * LDV + cbase + 2 * s + lh], Vj[l31 * LDV + cbase + 2 * s + lh], dp, 0, 0, 0);
    // ---- 2. exchange: every wave needs the full S and dP tiles
    {
      float* ex = EX + wid * 2048 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ex[r * 64] = sp[r];
        ex[1024 + r * 64] = dp[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += EX[w * 2048 + r * 64 + lane];
        b += EX[w * 2048 + 1024 + r * 64 + lane];
      }
      sp[r] = a;
      dp[r] = b;
    }
    // ---- 3. P and dS (register r of half lh is query (r&3) + 8*(r>>2) + 4*lh)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float p = __expf(sp[r] - lse_i[qi]);
      sp[r] = p;
      dp[r] = p * (dp[r] - del_i[qi]);
    }
    // ---- 4. dV^T[c][key] += gO^T[c][q] P[q][key] for this wave's channels
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int qi = (s & 3) + 8 * (s >> 2) + 4 * lh;
#pragma unroll
      for (int c = 0; c < NCW; ++c)
        acc_dv[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOi[qi * LDV + cbase + c * 32 + l31], sp[s], acc_dv[c], 0, 0, 0);
    }
    // ---- 5. key side (dK^T[d][key] += Q^T[d][q] dS[q][key]) or query side (dQ[q][d] = dS[q][key] K[key][d])
    if (dk_role) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int qi = (s & 3) + 8 * (s >> 2) + 4 * lh;
        acc_dk = __builtin_amdgcn_mfma_f32_32x32x2f32(Qi[qi * LDQ + dt * 32 + l31], dp[s], acc_dk, 0, 0, 0);
      }
    } else if (dq_role) {
      float* t = dsT + (wid - 2) * (32 * 33);
#pragma unroll
      for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = dp[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote it; LDS operations of a wave complete in order
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(t[l31 * 33 + 2 * s + lh], Kj[(2 * s + lh) * LDQ + dt * 32 + l31], dq, 0, 0, 0);
      float* gqb = gq + ((int64_t)n * T + i0) * D + dt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(gqb + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * D, dq[r]);
    }
    __syncthreads();   // all reads of the current query tile are done
    lstore();
    __syncthreads();
  }

  // all query-side atomics of this workgroup have COMPLETED before its key-side ones start: where both touch one row (the last query
  // tile of the last key block) their order is then fixed, which the reproducible mode relies on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: dV rows (plain stores, this workgroup is their only writer) and the key-side dQ (atomics)
  {
    const int64_t row = (int64_t)n * T + j0 + l31;
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const int ch = cbase + c * 32;
      float* ob = (ch < C1) ? gv1 + row * C1 + ch : gv2 + row * C2 + (ch - C1);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) =
            make_float4(acc_dv[c][4 * g], acc_dv[c][4 * g + 1], acc_dv[c][4 * g + 2], acc_dv[c][4 * g + 3]);
    }
    if (dk_role) {
      float* gqb = gq + row * D + dt * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(gqb + (r & 3) + 8 * (r >> 2) + 4 * lh, acc_dk[r]);
    }
  }
}

// =====================================================================================================
// Backward, second structure (used when it fits): every wave owns its OWN block of 32 keys with ALL value channels and
// keeps that key block's V and K fragments in registers (the lane-constant B operands of the score and dP products), so
//  * nothing is exchanged between waves before the softmax backward (the first structure above exchanges partial tiles),
//  * one query tile staged in LDS feeds four key blocks (4x fewer L2 reads of gO / Q),
//  * the query-side dQ tiles of the four key blocks are summed through LDS first: 4x fewer fp32 atomics
//    (rocprofv3 WRITE_SIZE showed 17 GB of atomic traffic per launch for the first structure at T = 16384).
// One wave per SIMD, ~420 of the 512 registers; 352 MFMAs per wave between barriers.
// =====================================================================================================
template <int D, int NCT>
__global__ void __launch_bounds__(256, 1) attn_bwd2_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                           const float* __restrict__ v2, const float* __restrict__ g1,
                                                           const float* __restrict__ g2, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, float* __restrict__ gv1,
                                                           float* __restrict__ gv2, float* __restrict__ gq, int T, int C1, int C2, int kb0) {
  constexpr int CT = NCT * 32, LDV = CT + 1, LDQ = D + 1, NDT = D / 32;
  constexpr int NQL = (8 * D) / 256 > 0 ? (8 * D) / 256 : 1, NVL = (8 * CT) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dOi = smem;                   // [32][LDV]  query tile of the upstream gradient
  float* Qi = dOi + 32 * LDV;          // [32][LDQ]
  float* lse_i = Qi + 32 * LDQ;        // [32]
  float* del_i = lse_i + 32;           // [32]
  float* Kw = del_i + 32;              // [4][32][LDQ]   each wave's key block (B operand of the query-side product)
  float* dsT = Kw + 4 * 32 * LDQ;      // [4][32][33]    private transposed dS tiles
  float* RB = dsT + 4 * 32 * 33;       // [4][NDT][16][64] query-side partial tiles, summed over the four key blocks

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y, j0 = ((int)blockIdx.x + kb0) * 128 + wid * 32;   // this wave's keys (kb0: key-block offset of this launch)
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;
  const float* g1b = g1 + (int64_t)n * T * C1;
  const float* g2b = g2 ? g2 + (int64_t)n * T * C2 : nullptr;
  const float* lseb = lse + (int64_t)n * T;
  const float* delb = delta + (int64_t)n * T;

  float* kw = Kw + wid * 32 * LDQ;
  for (int f = lane; f < 8 * D; f += 64) {
    const int key = f / (D / 4), dd = (f % (D / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(qb + (int64_t)(j0 + key) * D + dd);
    float* d = kw + key * LDQ + dd;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  float kfrag[D / 2], vfrag[CT / 2];   // K[key = l31][2s + lh], V[key = l31][2s + lh]
#pragma unroll
  for (int s = 0; s < D / 2; ++s) kfrag[s] = qb[(int64_t)(j0 + l31) * D + 2 * s + lh];
#pragma unroll
  for (int s = 0; s < CT / 2; ++s) {
    const int c = 2 * s + lh;
    vfrag[s] = (c < C1) ? v1b[(int64_t)(j0 + l31) * C1 + c] : v2b[(int64_t)(j0 + l31) * C2 + (c - C1)];
  }

  f32x16 acc_dv[NCT], acc_dk[NDT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dv[c][r] = 0.f;
#pragma unroll
  for (int c = 0; c < NDT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dk[c][r] = 0.f;

  float4 rq[NQL], rg[NVL];
  float rl = 0.f, rd = 0.f;
  auto gload = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (D / 4), dd = (f % (D / 4)) * 4;
      rq[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(i0 + row) * D + dd) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (CT / 4), c = (f % (CT / 4)) * 4;
      rg[i] = (c < C1) ? *reinterpret_cast<const float4*>(g1b + (int64_t)(i0 + row) * C1 + c)
                       : *reinterpret_cast<const float4*>(g2b + (int64_t)(i0 + row) * C2 + (c - C1));
    }
    rl = lseb[i0 + (tid & 31)];
    rd = delb[i0 + (tid & 31)];
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
      if (f < 8 * D) {
        float* d = Qi + (f / (D / 4)) * LDQ + (f % (D / 4)) * 4;
        d[0] = rq[i].x; d[1] = rq[i].y; d[2] = rq[i].z; d[3] = rq[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 256 * i;
      float* d = dOi + (f / (CT / 4)) * LDV + (f % (CT / 4)) * 4;
      d[0] = rg[i].x; d[1] = rg[i].y; d[2] = rg[i].z; d[3] = rg[i].w;
    }
    if (tid < 32) {
      lse_i[tid] = rl;
      del_i[tid] = rd;
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  float* t = dsT + wid * (32 * 33);
  for (int i0 = 0; i0 < T; i0 += 32) {
    gload(i0 + 32 < T ? i0 + 32 : i0);
    // ---- S[q][key], dP[q][key]: lane = key, registers = queries
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sp[r] = 0.f;
      dp[r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < D / 2; ++s) sp = __builtin_amdgcn_mfma_f32_32x32x2f32(Qi[l31 * LDQ + 2 * s + lh], kfrag[s], sp, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < CT / 2; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dOi[l31 * LDV + 2 * s + lh], vfrag[s], dp, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float p = __expf(sp[r] - lse_i[qi]);
      sp[r] = p;
      dp[r] = p * (dp[r] - del_i[qi]);
    }
    // ---- dV^T[c][key] += gO^T[c][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int qi = (s & 3) + 8 * (s >> 2) + 4 * lh;
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc_dv[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOi[qi * LDV + c * 32 + l31], sp[s], acc_dv[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NDT; ++c) acc_dk[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qi[qi * LDQ + c * 32 + l31], dp[s], acc_dk[c], 0, 0, 0);
    }
    // ---- query side: dQ[q][d] = dS[q][key] K[key][d] for this wave's keys, through the private transposed tile
#pragma unroll
    for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = dp[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < NDT; ++c) {
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(t[l31 * 33 + 2 * s + lh], kw[(2 * s + lh) * LDQ + c * 32 + l31], dq, 0, 0, 0);
      float* rb = RB + ((wid * NDT + c) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) rb[r * 64] = dq[r];
    }
    __syncthreads();   // every wave is done with the query tile; RB holds the four partial dQ tiles
    // sum the partials: NDT*16 register-rows in all, wave w takes rows [w*NDT*4, (w+1)*NDT*4) and issues the atomics
#pragma unroll
    for (int rr = 0; rr < NDT * 4; ++rr) {
      const int row = wid * NDT * 4 + rr;       // = c*16 + r
      const int c = row >> 4, r = row & 15;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += RB[((w * NDT + c) * 16 + r) * 64 + lane];
      atomicAdd(gq + ((int64_t)n * T + i0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + c * 32 + l31, s);
    }
    lstore();
    __syncthreads();
  }

  // all query-side atomics of this workgroup have COMPLETED before its key-side ones start: where both touch one row (the last query
  // tile of the last key block) their order is then fixed, which the reproducible mode relies on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: dV rows of this wave's keys (plain stores) and the key-side dQ (atomics)
  const int64_t row = (int64_t)n * T + j0 + l31;
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int ch = c * 32;
    float* ob = (ch < C1) ? gv1 + row * C1 + ch : gv2 + row * C2 + (ch - C1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) = make_float4(acc_dv[c][4 * g], acc_dv[c][4 * g + 1], acc_dv[c][4 * g + 2], acc_dv[c][4 * g + 3]);
  }
#pragma unroll
  for (int c = 0; c < NDT; ++c) {
    float* gqb = gq + row * D + c * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(gqb + (r & 3) + 8 * (r >> 2) + 4 * lh, acc_dk[c][r]);
  }
}

// =====================================================================================================
// Backward, second structure, on the bf16 matrix pipe (x6.h).  Same roles as attn_bwd2_kernel (a wave owns 32 keys with all value
// channels; the query tile staged in LDS feeds the four key blocks of the workgroup), every product as six bf16 MFMAs of K = 16:
//  * the query tile (Q_i, gO_i) is cut into its three bf16 pieces ONCE, on the way from the staging registers into LDS:
//      gO pieces [32 q][CT], 16-byte chunks XOR-swizzled (chunk ^ (((q & 3) << 2) | ((q >> 2) & 3)) inside each 256-byte window) so
//      that the SAME image serves the row fragments of dP = gO V^T (ds_read_b128) and the transposed fragments of dV^T = gO^T P
//      (ds_read_b64_tr_b16), both conflict-free;
//      Q pieces  [32 q][D], 192-byte rows: row fragments for S, transposed fragments for dK^T = Q^T dS;
//  * the wave's own K / V rows stay fp32 in registers, laid out as B fragments (8 consecutive reduction indices per lane), and
//    are split on the fly next to the MFMAs that consume them: three-piece copies of V would need 192 registers on top of the
//    160 accumulator registers;
//  * P and dS (lane = key, registers = queries) are split in registers: they are the B operands of dV^T and dK^T as they stand;
//    for the query-side product dQ = dS K the three dS pieces go through a private [key][q] LDS image and come back transposed
//    (ds_read_b64_tr_b16) as A fragments;
//  * the query-side tiles of the four key blocks are summed in LDS (ds_add_f32) before the global atomics, as before.
// Per query tile and wave: 264 bf16 MFMAs (8 448 matrix-pipe cycles; the fp32 kernel: 352 MFMAs, 22 528 cycles).
// =====================================================================================================
// DVLO, DVN: the value-channel tiles (32 channels each) whose dV this launch accumulates; FULL: also dP, dS, dK and both dQ terms.
// With all 8 channel tiles in one launch the 128 dV + 32 dK accumulator registers leave no room for the V pieces (192 registers): V stays
// fp32 and is split per tile (700 VALU instructions of ~2 400).  Two launches -- FULL with tiles 0..3, then a light one (S, P and dV of tiles
// 4..7 only: no V, no dP, no atomics) -- cost 288 instead of 264 MFMAs per tile pair, but the first then holds 96 accumulator registers and
// the optimiser keeps the V pieces across query tiles.
#ifdef FMI_ATT_STAMP  // diagnostic build only (tools/bench_tools/build_flags.sh attstamp -DFMI_ATT_STAMP): cycle stamps of the phases of ONE query tile
__device__ unsigned long long fmi_att_stamps[16];
extern "C" int fmi_debug_attn_stamps(unsigned long long* host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(fmi_att_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? FMI_OK : FMI_ERR_LAUNCH;
}
#define ATT_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (blockIdx.x == 0 && blockIdx.y == 0 && i0 == 32 * 100 && wid == 0 && lane == 0 && FULL)          \
      fmi_att_stamps[i] = __builtin_amdgcn_s_memtime();                                                \
  } while (0)
#else
#define ATT_STAMP(i)
#endif
template <int D, int NCT, int DVLO, int DVN, bool FULL>
__global__ void __launch_bounds__(256, 1) attn_bwd2_x6_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                              const float* __restrict__ v2, const float* __restrict__ g1,
                                                              const float* __restrict__ g2, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, float* __restrict__ gv1,
                                                              float* __restrict__ gv2, float* __restrict__ gq, int T, int C1, int C2, int kb0) {
  constexpr int CT = NCT * 32, LDQ = D + 1, NDT = D / 32;
  constexpr int GLO = FULL ? 0 : DVLO * 32, GN = FULL ? CT : DVN * 32;   // gO channels staged per query tile
  constexpr int HOIST_V = FULL && 2 * DVN <= NCT;                        // room for the V pieces in registers
  constexpr int NQL = (8 * D) / 256 > 0 ? (8 * D) / 256 : 1, NVL = (8 * GN) / 256;
  constexpr int GP = 2 * CT, GIMG = 32 * GP;   // gO piece image: row pitch, bytes per piece
  constexpr int QP = 192, QIMG = 32 * QP;      // Q piece image
  constexpr int TIMG = 32 * 64;                // dS^T piece image: [32 keys][32 q] bf16
  constexpr int KP = 2 * D + 16, KIMG = 32 * KP;  // key-block piece image: row pitch (conflict-free row reads), bytes per piece
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* Gs = smem_b;                                     // [3][32][GP]
  unsigned char* Qs = Gs + 3 * GIMG;                              // [3][32][QP]
  float* lse_i = reinterpret_cast<float*>(Qs + 3 * QIMG);         // [32]
  float* del_i = lse_i + 32;                                      // [32]
  unsigned char* Ks = reinterpret_cast<unsigned char*>(del_i + 32);  // [4][3][32][KP]  each wave's key block as bf16 pieces: B operand of S (row reads) and of dQ = dS K (transposed reads)
  unsigned char* Ts = Ks + 4 * 3 * KIMG;                          // [4][WSLOT]  per wave: the transposed dS pieces [3][32][64 B], then (same bytes) its query-side partial tiles
  // a wave's partial dQ tiles [NDT][16][64] overwrite its own dS^T image once its transposed reads are done; the other waves read them between
  // the two barriers that follow, and the image is rewritten only after the second one.  (LDS float atomics instead of partial tiles: ~250
  // cycles per wave instruction, measured.)
  constexpr int WSLOT = (NDT * 16 * 64 * 4 > 3 * TIMG) ? NDT * 16 * 64 * 4 : 3 * TIMG;
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem_b;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y, j0 = ((int)blockIdx.x + kb0) * 128 + wid * 32;   // this wave's keys (kb0: key-block offset of this launch)
  const float* qb = q + (int64_t)n * T * D;
  const float* v1b = v1 + (int64_t)n * T * C1;
  const float* v2b = v2 ? v2 + (int64_t)n * T * C2 : nullptr;
  const float* g1b = g1 + (int64_t)n * T * C1;
  const float* g2b = g2 ? g2 + (int64_t)n * T * C2 : nullptr;
  const float* lseb = lse + (int64_t)n * T;
  const float* delb = delta + (int64_t)n * T;

  unsigned char* ks = Ks + wid * 3 * KIMG;
  for (int f = lane; f < 8 * D; f += 64) {
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
  // B fragments of this wave's keys, V[key = l31][16 kk + 8 lh + j], in registers (the K fragments are read from the piece image per tile)
  float vfrag[FULL ? CT / 16 : 1][8];
  if constexpr (FULL) {
#pragma unroll
    for (int kk = 0; kk < CT / 16; ++kk) {
      const int c = 16 * kk + 8 * lh;
      const float* src = (c < C1) ? v1b + (int64_t)(j0 + l31) * C1 + c : v2b + (int64_t)(j0 + l31) * C2 + (c - C1);
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      vfrag[kk][0] = a.x, vfrag[kk][1] = a.y, vfrag[kk][2] = a.z, vfrag[kk][3] = a.w;
      vfrag[kk][4] = b.x, vfrag[kk][5] = b.y, vfrag[kk][6] = b.z, vfrag[kk][7] = b.w;
    }
  }

  f32x16 acc_dv[DVN], acc_dk[NDT];
#pragma unroll
  for (int c = 0; c < DVN; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dv[c][r] = 0.f;
#pragma unroll
  for (int c = 0; c < NDT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_dk[c][r] = 0.f;

  float4 rq[NQL], rg[NVL];
  float rl = 0.f, rd = 0.f;
  auto gload = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (D / 4), dd = (f % (D / 4)) * 4;
      rq[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(i0 + row) * D + dd) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (GN / 4), c = GLO + (f % (GN / 4)) * 4;
      rg[i] = (c < C1) ? *reinterpret_cast<const float4*>(g1b + (int64_t)(i0 + row) * C1 + c)
                       : *reinterpret_cast<const float4*>(g2b + (int64_t)(i0 + row) * C2 + (c - C1));
    }
    rl = lseb[i0 + (tid & 31)];
    if constexpr (FULL) rd = delb[i0 + (tid & 31)];
  };
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int f = tid + 256 * i;
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
      const int f = tid + 256 * i;
      const int row = f / (GN / 4), c4 = GLO / 4 + f % (GN / 4), ch = c4 >> 1;
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
  auto tr2 = [&](uint32_t a0, uint32_t a1) {  // two transposed 4 x 16 blocks -> one 8-element fragment
    union {
      s16x4_t h[2];
      bf16x8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a1);
    // the return data must land in an architectural VGPR: with 160 accumulator registers the allocator also hands out AGPRs as LDS
    // destinations, and an LDS read into an AGPR stalls the LDS return path (SQ_LDS_DATA_FIFO_FULL 1e10 cycles: 38.7 -> 18.9 ms)
    asm volatile("" : "+v"(u.v));
    return u.v;
  };
  // the same read with the pin left to the caller (pin3 below): a pin is also the point where the wave WAITS for the data, so a fragment
  // that is read one step ahead gets pinned only after the MFMAs of the current step have been issued
  auto tr2_nowait = [&](uint32_t a0, uint32_t a1) {
    union {
      s16x4_t h[2];
      bf16x8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(uintptr_t)a1);
    return u.v;
  };
  auto pin3 = [&](bf16x8_t (&a)[3]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); };
  // per-lane address parts
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, gb = (lane >> 4) & 1;
  const uint32_t g_row = lds0 + (uint32_t)(l31 * GP);                  // row fragments of gO: + piece, + swizzled chunk
  const int g_rx = (lh ^ swz(l31)) & 15;                               // chunk (2 kk + lh) ^ swz(row) = (2 kk) ^ g_rx on the low four bits
  const uint32_t g_tr = lds0 + (uint32_t)((4 * lh + tq) * GP + 8 * (tp & 1));   // transposed fragments of gO: query rows 16 s + 4 lh + tq (+ 8)
  const int g_tx = (2 * gb + (tp >> 1)) ^ ((tq << 2) | lh);            // chunk 4 ct + 2 gb + (tp >> 1), swizzle of that row ((+8: ^ 2)
  const uint32_t q_row = lds0 + (uint32_t)(3 * GIMG + l31 * QP + 16 * lh);
  const uint32_t q_tr = lds0 + (uint32_t)(3 * GIMG + (4 * lh + tq) * QP + (16 * gb + 4 * tp) * 2);
  const uint32_t k_base = lds0 + (uint32_t)(Ks - smem_b) + (uint32_t)(wid * 3 * KIMG);
  const uint32_t k_row = k_base + (uint32_t)(l31 * KP + 16 * lh);                              // row fragments: key l31, + piece, + 32 kk
  const uint32_t k_tr = k_base + (uint32_t)((8 * lh + tq) * KP + (16 * gb + 4 * tp) * 2);      // transposed: key rows 16 s + 8 lh + tq (+ 4), + 64 c
  const uint32_t t_base = (uint32_t)(Ts - smem_b) + (uint32_t)(wid * WSLOT);
  float* RBw = reinterpret_cast<float*>(Ts + wid * WSLOT);
  // dS^T image: row = key (64 bytes = 8 slots of four queries); slot (2 g + lh) of key k is stored at slot ^ ((k >> 1) & 7): the 16 keys
  // one ds_write_b64 group covers then fall on 16 different bank pairs (unswizzled: two)
  const uint32_t t_wr = lds0 + t_base + (uint32_t)(l31 * 64);                    // + piece, + 8 * ((2 g + lh) ^ t_wx)
  const int t_wx = (l31 >> 1) & 7;
  const uint32_t t_tr = lds0 + t_base + (uint32_t)((8 * lh + tq) * 64);          // key rows 16 s + 8 lh + tq (+ 4): + 8 * ((4 gb + tp) ^ swizzle of that row)
  const int t_rx = (4 * lh) | (tq >> 1);                                         // ((row >> 1) & 7) of the first row; the row 4 further: + 2

  gload(0);
  lstore();
  __syncthreads();
  for (int i0 = 0; i0 < T; i0 += 32) {
    ATT_STAMP(0);
    // ---- S[q][key], dP[q][key]: lane = key, registers = queries
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sp[r] = 0.f;
      dp[r] = 0.f;
    }
    {  // fragments of step kk + 1 are read before the MFMAs of step kk are issued
      auto s_frag = [&](int kk, bf16x8_t (&a)[3], bf16x8_t (&b)[3]) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a[pc] = *(lp8)(uintptr_t)(q_row + (uint32_t)(pc * QIMG + kk * 32));
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[pc] = *(lp8)(uintptr_t)(k_row + (uint32_t)(pc * KIMG + kk * 32));
      };
      bf16x8_t a[3], b[3], an[3], bn[3];
      s_frag(0, a, b);
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk) {
        if (kk + 1 < D / 16) s_frag(kk + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        sp = mfma_x6(a, b, sp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          a[pc] = an[pc];
          b[pc] = bn[pc];
        }
      }
    }
    ATT_STAMP(1);
    // The swizzled chunk offsets of the fragment reads are two instructions each.  Computed from the plain lane constants they are
    // loop-invariant, and the optimiser keeps all of them (one per step: ~48 registers) across the tile loop -- parked in AGPRs and
    // fetched with v_accvgpr_read at best, spilled to scratch and reloaded in front of the read at worst.  The asm makes the lane
    // constant opaque at each use, so the offset is recomputed where it is needed.
    auto gt_frag = [&](int s, int c, bf16x8_t (&a)[3]) {  // gO^T fragment: channels 32 c .., queries 16 s ..
      int tx = g_tx;
      asm volatile("" : "+v"(tx));
      const uint32_t ch0 = (uint32_t)(16 * (((4 * c) & 15) ^ tx) + 16 * ((4 * c) & ~15));
      const uint32_t ch1 = (uint32_t)(16 * (((4 * c) & 15) ^ tx ^ 2) + 16 * ((4 * c) & ~15));
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        a[pc] = tr2_nowait(g_tr + (uint32_t)(pc * GIMG + 16 * s * GP) + ch0, g_tr + (uint32_t)(pc * GIMG + (16 * s + 8) * GP) + ch1);
    };
    if constexpr (FULL) {
      // ---- P first (it needs S only), then ONE loop over the channel steps that carries both products whose reduction runs over the
      // channels / whose output is channels: dP[q][key] += gO[q][c] V[key][c] (step kk: 16 channels; the V pieces are split per step,
      // ~45 VALU instructions) and dV^T[c][key] += gO^T[c][q] P[q][key] (step j = kk: one 32-channel tile of one 16-query half; no VALU).
      // Alone, the dP steps are bound by the split (an MFMA gap hides about five vector instructions: 309 cycles a step against 192 of
      // MFMA) while the dV steps leave their gaps empty; together a step has 12 MFMAs for the same 45 instructions.  Every fragment of
      // step kk + 1 is read / split before the MFMAs of step kk are issued and waited for (pinned) after them.
#pragma unroll
      for (int r = 0; r < 16; ++r) sp[r] = __expf(sp[r] - lse_i[(r & 3) + 8 * (r >> 2) + 4 * lh]);
      bf16x8_t pp[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float f[8] = {sp[8 * s], sp[8 * s + 1], sp[8 * s + 2], sp[8 * s + 3], sp[8 * s + 4], sp[8 * s + 5], sp[8 * s + 6], sp[8 * s + 7]};
        split3_bf16(f, pp[s]);
      }
      auto g_frag = [&](int kk, bf16x8_t (&a)[3]) {
        int rx = g_rx;
        asm volatile("" : "+v"(rx));
        const uint32_t ad = g_row + (uint32_t)(16 * (((2 * kk) & 15) ^ rx) + 16 * ((2 * kk) & ~15));
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a[pc] = *(lp8)(uintptr_t)(ad + (uint32_t)(pc * GIMG));
      };
      // half h of the split of V step kk: values 4 h .. 4 h + 3 -> words 2 h, 2 h + 1 of the three piece fragments
      auto v_split_half = [&](int kk, int h, u32x4_t (&w)[3]) {
#pragma unroll
        for (int e = 2 * h; e < 2 * h + 2; ++e) {
          uint32_t w0, w1, w2;
          split3_pair(vfrag[kk][2 * e], vfrag[kk][2 * e + 1], w0, w1, w2);
          w[0][e] = w0, w[1][e] = w1, w[2][e] = w2;
        }
      };
      // The V values must not look loop-invariant to the optimiser: it would hoist all 192 piece registers out of the tile loop and
      // spill.  An empty asm that "modifies" them is free only while the value stays in the register class the constraint names: the
      // accumulators take 192 of the 256 AGPRs, so half of the V fragments live in VGPRs ("+v") and half in the remaining AGPRs ("+a";
      // with "+v" on all of them each use cost a v_accvgpr_read AND a v_accvgpr_write back)
      auto v_touch = [&](int kk) {
        if constexpr (!HOIST_V) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (kk < (CT / 16) / 2)
              asm volatile("" : "+v"(vfrag[kk][j]));
            else
              asm volatile("" : "+a"(vfrag[kk][j]));
          }
        }
      };
      auto ds_make = [&](int s, bf16x8_t (&ds)[3]) {
        const float e[8] = {dp[8 * s], dp[8 * s + 1], dp[8 * s + 2], dp[8 * s + 3], dp[8 * s + 4], dp[8 * s + 5], dp[8 * s + 6], dp[8 * s + 7]};
        split3_bf16(e, ds);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {  // registers 8 s .. 8 s + 3 are queries 16 s + 4 lh .., registers 8 s + 4 .. + 7 queries 16 s + 8 + 4 lh ..
          typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(3))) u32x2_t* lpu2;
          const u32x4_t w = __builtin_bit_cast(u32x4_t, ds[pc]);
          *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s) | lh) ^ t_wx))) = u32x2_t{w[0], w[1]};      // g = 2 s
          *(lpu2)(uintptr_t)(t_wr + (uint32_t)(pc * TIMG + 8 * (((4 * s + 2) | lh) ^ t_wx))) = u32x2_t{w[2], w[3]};  // g = 2 s + 1
        }
      };
      auto qt_frag = [&](int s, bf16x8_t (&aq)[NDT][3]) {
#pragma unroll
        for (int c = 0; c < NDT; ++c)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            aq[c][pc] = tr2_nowait(q_tr + (uint32_t)(pc * QIMG + 16 * s * QP + 64 * c), q_tr + (uint32_t)(pc * QIMG + (16 * s + 8) * QP + 64 * c));
      };
      bf16x8_t aq0[NDT][3], aq1[NDT][3], ds0[3], ds1[3];
      constexpr int NDP = CT / 16, NDV = 2 * DVN;   // NDV <= NDP: the dV step of the tile (s, c) = (j / DVN, j % DVN) rides on dP step j
      static_assert(NDV <= NDP, "dV steps ride on the dP steps");
      // A step has two halves of six MFMAs.  First half: dP with the fragments a (gO rows) and b (V pieces) made during the step before;
      // the gO^T fragment t of this step's dV tile is read at its start.  Second half: dV with t; a is dead by then and receives the gO
      // rows of step kk + 1.  The split of V step kk + 1 is spread over both halves, four vector instructions behind each MFMA (an MFMA
      // gap hides vector issue only while it stays within its 32 cycles: about five instructions; left alone the scheduler puts 9 - 16
      // of them between two MFMAs and then issues five MFMAs back to back -- 537 cycles a step against 384 of MFMA, stamps)
      bf16x8_t a[3], b[3], t[3];
      u32x4_t wn[3];
      g_frag(0, a);
      v_touch(0);
      v_split_half(0, 0, wn);
      v_split_half(0, 1, wn);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) b[pc] = __builtin_bit_cast(bf16x8_t, wn[pc]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < NDP; ++kk) {
        if (kk < NDV) gt_frag(kk / DVN, DVLO + kk % DVN, t);
        if (kk + 1 < NDP) {
          v_touch(kk + 1);
          v_split_half(kk + 1, 0, wn);
        }
        dp = mfma_x6(a, b, dp);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kk < NDV) pin3(t);
        if (kk + 1 < NDP) {
          g_frag(kk + 1, a);
          v_split_half(kk + 1, 1, wn);
        }
        if (kk < NDV) acc_dv[kk < NDV ? kk % DVN : 0] = mfma_x6(t, pp[kk < NDV ? kk / DVN : 0], acc_dv[kk < NDV ? kk % DVN : 0]);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[pc] = __builtin_bit_cast(bf16x8_t, wn[pc]);
      }
    ATT_STAMP(2);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = sp[r] * (dp[r] - del_i[(r & 3) + 8 * (r >> 2) + 4 * lh]);
    ATT_STAMP(3);
      qt_frag(0, aq0);
      ds_make(0, ds0);
      // ---- dK^T[d][key] += Q^T[d][q] dS[q][key];  dS pieces -> private transposed image (A operand of dQ below).
      // The pieces of the second 16-query half are split and stored behind the MFMAs of the first half, and the next query tile's
      // global loads (address arithmetic: ~70 vector instructions) are issued behind the MFMAs of the second half: their 40 staging
      // registers are live from here on only -- across the dP / dV loop they would not fit (82 registers went to scratch)
#pragma unroll
      for (int c = 0; c < NDT; ++c) pin3(aq0[c]);
      __builtin_amdgcn_sched_barrier(0);
      qt_frag(1, aq1);
      ds_make(1, ds1);
#pragma unroll
      for (int c = 0; c < NDT; ++c) acc_dk[c] = mfma_x6(aq0[c], ds0, acc_dk[c]);
      __builtin_amdgcn_sched_group_barrier(0x100, 6 * NDT, 0);
#pragma unroll
      for (int g = 0; g < 6 * NDT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NDT; ++c) pin3(aq1[c]);
    ATT_STAMP(4);
      gload(i0 + 32 < T ? i0 + 32 : i0);
#pragma unroll
      for (int c = 0; c < NDT; ++c) acc_dk[c] = mfma_x6(aq1[c], ds1, acc_dk[c]);
#pragma unroll
      for (int g = 0; g < 6 * NDT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // ---- second pass over the other channels: P, then dV^T[c][key] += gO^T[c][q] P[q][key] alone
    ATT_STAMP(2);
#pragma unroll
      for (int r = 0; r < 16; ++r) sp[r] = __expf(sp[r] - lse_i[(r & 3) + 8 * (r >> 2) + 4 * lh]);
    ATT_STAMP(3);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t pp[3];
        const float f[8] = {sp[8 * s], sp[8 * s + 1], sp[8 * s + 2], sp[8 * s + 3], sp[8 * s + 4], sp[8 * s + 5], sp[8 * s + 6], sp[8 * s + 7]};
        split3_bf16(f, pp);
        bf16x8_t a[3], an[3];
        gt_frag(s, DVLO, a);
        pin3(a);
#pragma unroll
        for (int j = 0; j < DVN; ++j) {  // fragment j + 1 is read before the MFMAs of fragment j are issued and pinned after them
          if (j + 1 < DVN) gt_frag(s, DVLO + j + 1, an);
          __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks the reads below the MFMAs, right in front of their wait
          acc_dv[j] = mfma_x6(a, pp, acc_dv[j]);
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < DVN) pin3(an);
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) a[pc] = an[pc];
        }
      }
    }
    if constexpr (!FULL) {
    ATT_STAMP(4);
      __builtin_amdgcn_sched_barrier(0);
      gload(i0 + 32 < T ? i0 + 32 : i0);
    }
    if constexpr (FULL) {
      // ---- query side: dQ[q][d] = dS[q][key] K[key][d] for this wave's keys
    ATT_STAMP(5);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transposed image was written by this wave; a wave's LDS operations complete in order
      f32x16 dqs[NDT];
      auto dq_frag = [&](int j, bf16x8_t (&a)[3], bf16x8_t (&b)[3]) {  // step j = (c, s)
        const int c = j >> 1, s = j & 1;
  #pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          a[pc] = tr2_nowait(t_tr + (uint32_t)(pc * TIMG + 16 * s * 64 + 8 * ((4 * gb + tp) ^ t_rx)),
                             t_tr + (uint32_t)(pc * TIMG + (16 * s + 4) * 64 + 8 * ((4 * gb + tp) ^ (t_rx + 2))));
        }
  #pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          b[pc] = tr2_nowait(k_tr + (uint32_t)(pc * KIMG + 16 * s * KP + 64 * c), k_tr + (uint32_t)(pc * KIMG + (16 * s + 4) * KP + 64 * c));
      };
      bf16x8_t a[3], b[3], an[3], bn[3];
      dq_frag(0, a, b);
      pin3(a);
      pin3(b);
  #pragma unroll
      for (int j = 0; j < 2 * NDT; ++j) {  // same pipeline as above: the fragments of step j + 1 are in flight while step j multiplies
        if (j + 1 < 2 * NDT) dq_frag(j + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        if ((j & 1) == 0) {
  #pragma unroll
          for (int r = 0; r < 16; ++r) dqs[j >> 1][r] = 0.f;
        }
        dqs[j >> 1] = mfma_x6(a, b, dqs[j >> 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < 2 * NDT) {
          pin3(an);
          pin3(bn);
        }
  #pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          a[pc] = an[pc];
          b[pc] = bn[pc];
        }
      }
    ATT_STAMP(6);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's transposed reads of its dS^T image are done: the partial tiles may overwrite it
  #pragma unroll
      for (int c = 0; c < NDT; ++c)
  #pragma unroll
        for (int r = 0; r < 16; ++r) RBw[(c * 16 + r) * 64 + lane] = dqs[c][r];
    ATT_STAMP(7);
      __syncthreads();   // every wave is done with the query tile; the four slots hold the partial dQ tiles
    ATT_STAMP(8);
      // sum the partials: NDT*16 register-rows in all, wave w takes rows [w*NDT*4, (w+1)*NDT*4) and issues the atomics.  All the partials
      // are read first; the next query tile's pieces are split and stored while they are in flight (every wave is past the barrier, so
      // nobody reads the gO / Q images any more), then the sums and the atomics follow
      float part[NDT * 4][4];
  #pragma unroll
      for (int rr = 0; rr < NDT * 4; ++rr)
  #pragma unroll
        for (int w = 0; w < 4; ++w) part[rr][w] = reinterpret_cast<const float*>(Ts + w * WSLOT)[(wid * NDT * 4 + rr) * 64 + lane];
    ATT_STAMP(9);
      lstore();
    ATT_STAMP(10);
  #pragma unroll
      for (int rr = 0; rr < NDT * 4; ++rr) {
        const int row = wid * NDT * 4 + rr;       // = c*16 + r
        const int c = row >> 4, r = row & 15;
        float sum = 0.f;
  #pragma unroll
        for (int w = 0; w < 4; ++w) sum += part[rr][w];
        atomicAdd(gq + ((int64_t)n * T + i0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * D + c * 32 + l31, sum);
      }
    } else {
      __syncthreads();   // every wave is done with the query tile
      lstore();
    }
    __syncthreads();
    ATT_STAMP(11);
  }

  // all query-side atomics of this workgroup have COMPLETED before its key-side ones start: where both touch one row (the last query
  // tile of the last key block) their order is then fixed, which the reproducible mode relies on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: dV rows of this wave's keys (plain stores) and the key-side dQ (atomics)
  const int64_t row = (int64_t)n * T + j0 + l31;
#pragma unroll
  for (int c = 0; c < DVN; ++c) {
    const int ch = (DVLO + c) * 32;
    float* ob = (ch < C1) ? gv1 + row * C1 + ch : gv2 + row * C2 + (ch - C1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ob + 8 * g + 4 * lh) = make_float4(acc_dv[c][4 * g], acc_dv[c][4 * g + 1], acc_dv[c][4 * g + 2], acc_dv[c][4 * g + 3]);
  }
#pragma unroll
  for (int c = 0; c < (FULL ? NDT : 0); ++c) {
    float* gqb = gq + row * D + c * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(gqb + (r & 3) + 8 * (r >> 2) + 4 * lh, acc_dk[c][r]);
  }
}

// delta[row] = sum_c a1[row][c] b1[row][c] (+ second pair): one wave per row
__global__ void __launch_bounds__(256) rowdot2_kernel(const float* __restrict__ a1, const float* __restrict__ b1, int C1,
                                                      const float* __restrict__ a2, const float* __restrict__ b2, int C2,
                                                      float* __restrict__ out, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C1; c += 64) s += a1[row * C1 + c] * b1[row * C1 + c];
  if (a2)
    for (int c = lane; c < C2; c += 64) s += a2[row * C2 + c] * b2[row * C2 + c];
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

extern "C" int fmi_attention_bwd_f32(const float* q, const float* v1, const float* v2, const float* o1, const float* o2,
                                     const float* go1, const float* go2, const float* lse, float* delta_scratch, float* gv1,
                                     float* gv2, float* gq_zeroed, int N, int T, int D, int C1, int C2, void* stream) {
  if (!q || !v1 || !o1 || !go1 || !lse || !delta_scratch || !gv1 || !gq_zeroed || N <= 0 || T <= 0 || C1 <= 0 || C2 < 0)
    return FMI_ERR_BAD_ARG;
  if (C2 > 0 && (!v2 || !o2 || !go2 || !gv2)) return FMI_ERR_BAD_ARG;
  if (T % 32 != 0 || C1 % 32 != 0 || C2 % 32 != 0 || N > 65535) return FMI_ERR_UNSUPPORTED;
  if ((((uintptr_t)q | (uintptr_t)v1 | (uintptr_t)v2 | (uintptr_t)go1 | (uintptr_t)go2 | (uintptr_t)gv1 | (uintptr_t)gv2) & 15) != 0)
    return FMI_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)N * T;
  hipLaunchKernelGGL(rowdot2_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, go1, o1, C1, C2 ? go2 : nullptr, o2, C2,
                     delta_scratch, rows);
  const int nct = (C1 + C2) / 32;
  const dim3 block(256);
  // reproducible mode: the query-side tiles of dQ receive one atomic contribution per KEY BLOCK; launching the key blocks one after the
  // other (stream order) fixes the order of those additions -- same kernels, grid.x = 1, key-block offset as an argument
  const bool det = fmi_det();
  static const int bwd_dbg = getenv("FMI_ATT_BWD") ? atoi(getenv("FMI_ATT_BWD")) : 0;  // debug: 1 / 2 force a structure
  const bool small = (int64_t)(T / 128) * N < 128;  // short sequences: the first structure has 4x the workgroups
  if (T % 128 == 0 && bwd_dbg != 1 && (bwd_dbg == 2 || !small)) {  // second structure: one key block per wave, fragments in registers, 4x fewer atomics
    const dim3 grid2(T / 128, N);
    auto lds2 = [](int d, int ct) {
      return sizeof(float) * (size_t)(32 * (ct + 1) + 32 * (d + 1) + 64 + 4 * 32 * (d + 1) + 4 * 32 * 33 + 4 * (d / 32) * 16 * 64);
    };
    auto lds2x = [](int d, int ct) {
      return (size_t)(3 * 32 * 2 * ct + 3 * 32 * 192 + 64 * 4 + 4 * 3 * 32 * (2 * d + 16) + 4 * ((d / 32) * 4096 > 6144 ? (d / 32) * 4096 : 6144));
    };
    // two launches (dV of channel tiles 0..3 with everything else, then a light S / P / dV launch for tiles 4..7) let the first keep the V
    // pieces in registers: measured 14.9 + 5.6 ms against 17.6 ms for the single launch -- off unless FMI_ATT_BWD_2PASS is set
    static const bool one_pass = getenv("FMI_ATT_BWD_2PASS") == nullptr;
#define ATTB2_X6(DD, NN, LO, CNT, FULLP)                                                                                 \
  do {                                                                                                                   \
    static fmi_attr_flags attr_setx{}; int attr_setx_dev;\
    if (fmi_attr_needed(attr_setx, attr_setx_dev)) {                                                                                                    \
      if (hipFuncSetAttribute((const void*)attn_bwd2_x6_kernel<DD, NN, LO, CNT, FULLP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds2x(DD, NN * 32)) != hipSuccess)                                                    \
        return FMI_ERR_LAUNCH;                                                                                           \
      fmi_attr_mark(attr_setx, attr_setx_dev);                                                                                                  \
    }                                                                                                                    \
    for (int kb = 0; kb < (det ? (int)grid2.x : 1); ++kb)                                                                \
      hipLaunchKernelGGL((attn_bwd2_x6_kernel<DD, NN, LO, CNT, FULLP>), det ? dim3(1, grid2.y) : grid2, block, lds2x(DD, NN * 32), st, q, v1, v2, go1, go2, lse, \
                         (const float*)delta_scratch, gv1, gv2, gq_zeroed, T, C1, C2, kb);                                \
  } while (0)
#define ATTB2_LAUNCH(DD, NN)                                                                                             \
  do {                                                                                                                   \
    if (FMI_X6) {                                                                                                        \
      if (NN == 8 && !one_pass) {                                                                                        \
        ATTB2_X6(DD, NN, 0, NN / 2, true);                                                                               \
        ATTB2_X6(DD, NN, NN / 2, NN / 2, false);                                                                         \
      } else {                                                                                                           \
        ATTB2_X6(DD, NN, 0, NN, true);                                                                                   \
      }                                                                                                                  \
      return fmi_launch_status();                                                                                        \
    }                                                                                                                    \
    static fmi_attr_flags attr_set2{}; int attr_set2_dev;\
    if (fmi_attr_needed(attr_set2, attr_set2_dev)) {                                                                                                    \
      if (hipFuncSetAttribute((const void*)attn_bwd2_kernel<DD, NN>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                              (int)lds2(DD, NN * 32)) != hipSuccess)                                                     \
        return FMI_ERR_LAUNCH;                                                                                           \
      fmi_attr_mark(attr_set2, attr_set2_dev);                                                                                                  \
    }                                                                                                                    \
    for (int kb = 0; kb < (det ? (int)grid2.x : 1); ++kb)                                                                \
      hipLaunchKernelGGL((attn_bwd2_kernel<DD, NN>), det ? dim3(1, grid2.y) : grid2, block, lds2(DD, NN * 32), st, q, v1, v2, go1, go2, lse, \
                         (const float*)delta_scratch, gv1, gv2, gq_zeroed, T, C1, C2, kb);                                \
    return fmi_launch_status();                                                                                          \
  } while (0)
    if (D == 64 && nct == 8) ATTB2_LAUNCH(64, 8);
    if (D == 32 && nct == 8) ATTB2_LAUNCH(32, 8);
    if (D == 32 && nct == 4) ATTB2_LAUNCH(32, 4);
    if (D == 64 && nct == 4) ATTB2_LAUNCH(64, 4);
#undef ATTB2_LAUNCH
#undef ATTB2_X6
  }
  const dim3 grid(T / 32, N);
  auto lds_bytes = [](int d, int ct) { return sizeof(float) * (size_t)(2 * 32 * (ct + 1) + 2 * 32 * (d + 1) + 64 + 4 * 2 * 1024 + 2 * 32 * 33); };
#define ATTB_LAUNCH(DD, NN)                                                                                              \
  do {                                                                                                                   \
    static fmi_attr_flags attr_set{}; int attr_set_dev;\
    if (fmi_attr_needed(attr_set, attr_set_dev)) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)attn_bwd_kernel<DD, NN>, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                              (int)lds_bytes(DD, NN * 32)) != hipSuccess)                                                \
        return FMI_ERR_LAUNCH;                                                                                           \
      fmi_attr_mark(attr_set, attr_set_dev);                                                                                                   \
    }                                                                                                                    \
    for (int kb = 0; kb < (det ? (int)grid.x : 1); ++kb)                                                                 \
      hipLaunchKernelGGL((attn_bwd_kernel<DD, NN>), det ? dim3(1, grid.y) : grid, block, lds_bytes(DD, NN * 32), st, q, v1, v2, go1, go2, lse, \
                         (const float*)delta_scratch, gv1, gv2, gq_zeroed, T, C1, C2, kb);                                \
  } while (0)
  if (D == 64 && nct == 8) ATTB_LAUNCH(64, 8);
  else if (D == 32 && nct == 8) ATTB_LAUNCH(32, 8);
  else if (D == 32 && nct == 4) ATTB_LAUNCH(32, 4);
  else if (D == 64 && nct == 4) ATTB_LAUNCH(64, 4);
  else return FMI_ERR_UNSUPPORTED;
#undef ATTB_LAUNCH
  return fmi_launch_status();
}

