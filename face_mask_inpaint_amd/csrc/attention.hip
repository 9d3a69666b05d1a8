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

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ATT_THR 20.0f

template <int D, int CT, int NKL, int NVL>
__device__ __forceinline__ void att_gload(float4 (&rk)[NKL], float4 (&rv)[NVL], const float* __restrict__ qb,
                                          const float* __restrict__ v1b, const float* __restrict__ v2b, int C1, int C2, int k0,
                                          int tid) {
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int f = tid + 256 * i;
    const int key = f / (D / 4), dq = f % (D / 4);
    rk[i] = (f < 8 * D) ? *reinterpret_cast<const float4*>(qb + (int64_t)(k0 + key) * D + dq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < NVL; ++i) {
    const int f = tid + 256 * i;
    const int key = f / (CT / 4), c = (f % (CT / 4)) * 4;
    rv[i] = (c < C1) ? *reinterpret_cast<const float4*>(v1b + (int64_t)(k0 + key) * C1 + c)
                     : *reinterpret_cast<const float4*>(v2b + (int64_t)(k0 + key) * C2 + (c - C1));
  }
}
template <int D, int CT, int NKL, int NVL, int LDK>
__device__ __forceinline__ void att_lstore(const float4 (&rk)[NKL], const float4 (&rv)[NVL], float* __restrict__ kt,
                                           float* __restrict__ vs, int tid) {
#pragma unroll
  for (int i = 0; i < NKL; ++i) {
    const int f = tid + 256 * i;
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
    const int f = tid + 256 * i;
    *reinterpret_cast<float4*>(vs + f * 4) = rv[i];  // f*4 = key*CT + c
  }
}

template <int D, int NCT>
__global__ void __launch_bounds__(256, 1) attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ v1,
                                                          const float* __restrict__ v2, float* __restrict__ o1,
                                                          float* __restrict__ o2, float* __restrict__ lse, int T, int C1,
                                                          int C2) {
  constexpr int CT = NCT * 32;        // total value channels
  constexpr int LDK = 33;             // Kt row pitch (floats)
  constexpr int KT_FLOATS = D * LDK, VS_FLOATS = 32 * CT;
  constexpr int NKL = (8 * D) / 256 > 0 ? (8 * D) / 256 : 1;   // float4 loads per thread for a K tile
  constexpr int NVL = (8 * CT) / 256;                          // float4 loads per thread for a V tile
  __shared__ __attribute__((aligned(16))) float lds[2 * (KT_FLOATS + VS_FLOATS)];
  float* Kt = lds;
  float* Vs = lds + 2 * KT_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;
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
  att_gload<D, CT, NKL, NVL>(rk, rv, qb, v1b, v2b, C1, C2, 0, tid);
  att_lstore<D, CT, NKL, NVL, LDK>(rk, rv, Kt, Vs, tid);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < T; k0 += 32) {
    // unconditional prefetch (the last iteration re-reads its own tile): a conditional one sends rk/rv to scratch
    att_gload<D, CT, NKL, NVL>(rk, rv, qb, v1b, v2b, C1, C2, k0 + 32 < T ? k0 + 32 : k0, tid);
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
    att_lstore<D, CT, NKL, NVL, LDK>(rk, rv, Kt + (buf ^ 1) * KT_FLOATS, Vs + (buf ^ 1) * VS_FLOATS, tid);
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
  if ((((uintptr_t)q | (uintptr_t)v1 | (uintptr_t)v2 | (uintptr_t)o1 | (uintptr_t)o2) & 15) != 0) return FMI_ERR_BAD_ARG;
  const int nct = (C1 + C2) / 32;
  const dim3 grid(T / 128, N), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ATT_LAUNCH(DD, NN) hipLaunchKernelGGL((attn_fwd_kernel<DD, NN>), grid, block, 0, st, q, v1, v2, o1, o2, lse, T, C1, C2)
  if (D == 64 && nct == 8) ATT_LAUNCH(64, 8);
  else if (D == 32 && nct == 8) ATT_LAUNCH(32, 8);
  else if (D == 32 && nct == 4) ATT_LAUNCH(32, 4);
  else if (D == 64 && nct == 4) ATT_LAUNCH(64, 4);
  else if (D == 16 && nct == 2) ATT_LAUNCH(16, 2);
  else if (D == 16 && nct == 4) ATT_LAUNCH(16, 4);
  else return FMI_ERR_UNSUPPORTED;
#undef ATT_LAUNCH
  return fmi_launch_status();
}
