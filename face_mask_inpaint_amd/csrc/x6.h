// fp32 products on the bf16 matrix pipe ("bf16x6"): shared by the GEMM / convolution core and the fused attention kernels.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- fp32 product on the bf16 matrix pipe ("bf16x6") ----
// gfx950 has no xf32 and runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate.  An fp32 value is cut exactly into three bf16 pieces
// x = x0 + x1 + x2 of 8 significant bits each; x y is then the six products x0 y0, x0 y1, x1 y0, x0 y2, x1 y1, x2 y0 (each exact in the
// fp32 accumulator) -- the three dropped ones are below 2^-25 |x y|, under one fp32 rounding.  6 bf16 MFMAs of K = 16 replace 8 fp32
// MFMAs of K = 2: 192 instead of 512 matrix-pipe cycles per 32 x 32 x 16 block; the split costs 5.5 VALU instructions per element.
#ifndef FMI_X6
#define FMI_X6 1  // 0: the v_mfma_f32_32x32x2_f32 path (kept for A/B timing; same results to fp32 rounding)
#endif
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// two fp32 values -> three words, word i = {piece i of a (low half), piece i of b (high half)}.  Round-to-nearest pieces (v_cvt_pk_bf16_f32):
// x0 = rn(x), x1 = rn(x - x0), x2 = x - x0 - x1 -- both differences are exact in fp32 and x2 has at most 8 significant bits, so
// x = x0 + x1 + x2 exactly, with |x1| <= 2^-9 |x| and |x2| <= 2^-18 |x|.  (Truncated pieces cost the same 11 instructions per pair but
// leave dropped cross terms four times larger and all of one sign.)  An infinite input yields NaN (inf - inf), not inf.  A piece that is a
// bf16 SUBNORMAL (|piece| < 2^-126: the third piece of |x| < 2^-100, the second of |x| < 2^-117) is flushed to zero by the matrix pipe: such
// operands degrade towards bf16 accuracy (tests/test_gpu_kernels.py::test_bf16x6_edge_values_on_the_matrix_pipe pins the behaviour).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  const bf16x2_t h = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& q0, uint32_t& q1, uint32_t& q2) {
  q0 = cvt_pk_bf16(a, b);
#if FMI_X6 == 2  // timing experiment: no split arithmetic (wrong results)
  q1 = __float_as_uint(a), q2 = __float_as_uint(b);
  return;
#endif
  const float ra = a - __uint_as_float(q0 << 16), rb = b - __uint_as_float(q0 & 0xFFFF0000u);
  q1 = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(q1 << 16), sb = rb - __uint_as_float(q1 & 0xFFFF0000u);
  q2 = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split3_bf16(const float (&f)[8], bf16x8_t (&p)[3]) {
  u32x4_t q0, q1, q2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t w0, w1, w2;
    split3_pair(f[2 * e], f[2 * e + 1], w0, w1, w2);
    q0[e] = w0, q1[e] = w1, q2[e] = w2;
  }
  p[0] = __builtin_bit_cast(bf16x8_t, q0);
  p[1] = __builtin_bit_cast(bf16x8_t, q1);
  p[2] = __builtin_bit_cast(bf16x8_t, q2);
}
__device__ __forceinline__ f32x16 mfma_x6(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16 c) {
#if FMI_X6 == 3  // timing experiment: one product per pair (wrong results)
  bf16x8_t a0 = a[0], b0 = b[0];
  for (int e = 0; e < 8; ++e) a0[e] += a[1][e] + a[2][e], b0[e] += b[1][e] + b[2][e];
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
#endif
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

