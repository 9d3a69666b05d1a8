// Dense batched GEMM entry point (see include/fmi_hip.h: fmi_gemm_f32).
#include "gemm_core.h"

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

#ifndef FMI_HOST_EMU
// C = beta*C + bias[n] over a strided [batch][M][N] region: prologue of the split-K path
__global__ void __launch_bounds__(256) c_prep_kernel(float* __restrict__ C, int M, int N, int64_t sc_m, int64_t sc_n,
                                                     int64_t sc_b, float beta, const float* __restrict__ bias, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % N);
    const int64_t r = i / N;
    const int m = (int)(r % M);
    const int64_t b = r / M;
    float* q = C + b * sc_b + (int64_t)m * sc_m + (int64_t)n * sc_n;
    float v = beta == 0.f ? 0.f : beta * *q;
    if (bias) v += bias[n];
    *q = v;
  }
}
#endif

template <class LA, class LB>
static int run(const LA& la, const LB& lb, float* C, int M, int N, int K, int64_t sc_m, int64_t sc_n, int64_t sc_b,
               int batch, float alpha, float beta, const float* bias, hipStream_t st) {
#ifndef FMI_HOST_EMU
  // Skinny outputs with a long reduction (attention P.V: 128 x 256 outputs, K = 16384) would occupy a handful of
  // CUs: split K over workgroups and let the partial tiles meet through fp32 atomics.
  const int64_t tiles = ceil_div64(M, M <= 64 ? 64 : 128) * ceil_div64(N, N <= 32 ? 32 : (N <= 64 ? 64 : 128)) * batch;
  // ... and so would the M = batch-size products of EqualLinear (16 x 512 x 512: 4 tiles walking 32 reduction tiles one after
  // the other, 22 us per launch, 66 launches in a decoder step): a few reduction tiles per workgroup there
  const bool tiny = M <= 32 && tiles < 32 && K >= 256 && K < 2048;
  if (((tiles < 192 && K >= 2048) || tiny) && !fmi_det()) {  // reproducible mode: no split reduction
    int64_t ks = tiny ? 128 / tiles : 768 / tiles;
    const int64_t kmin = tiny ? 64 : 512;
    if (ks > K / kmin) ks = K / kmin;
    if (ks * batch > 65535) ks = 65535 / batch;
    if (ks >= 2) {
      if (beta != 1.f || bias) {
        const int64_t total = (int64_t)batch * M * N;
        hipLaunchKernelGGL(c_prep_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, st, C, M, N, sc_m, sc_n, sc_b, beta, bias, total);
      }
      DenseEp ep{C, nullptr, sc_m, sc_n, sc_b, alpha, 0.f, 1};
      return launch_gemm(la, lb, ep, M, N, K, batch, (int)ks, st);
    }
  }
#endif
  DenseEp ep{C, bias, sc_m, sc_n, sc_b, alpha, beta, 0};
  return launch_gemm(la, lb, ep, M, N, K, batch, 1, st);
}

extern "C" int fmi_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sa_m, int64_t sa_k,
                            int64_t sb_k, int64_t sb_n, int64_t sc_m, int64_t sc_n, int batch, int64_t sa_b,
                            int64_t sb_b, int64_t sc_b, float alpha, float beta, const float* bias, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return FMI_ERR_BAD_ARG;
  if (sa_m != 1 && sa_k != 1) return FMI_ERR_UNSUPPORTED;
  if (sb_k != 1 && sb_n != 1) return FMI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool a_k = (sa_k == 1), b_k = (sb_k == 1);
  // float4 loads need 16-byte aligned rows in every batch
  const int64_t lda = a_k ? sa_m : sa_k, ldb = b_k ? sb_n : sb_k;
  const int avec = aligned16(A) && (lda % 4 == 0) && (sa_b % 4 == 0);
  const int bvec = aligned16(B) && (ldb % 4 == 0) && (sb_b % 4 == 0);
  if (a_k && b_k) {
    DenseK la{A, lda, sa_b, M, K, avec};
    DenseK lb{B, ldb, sb_b, N, K, bvec};
    return run(la, lb, C, M, N, K, sc_m, sc_n, sc_b, batch, alpha, beta, bias, st);
  } else if (a_k && !b_k) {
    DenseK la{A, lda, sa_b, M, K, avec};
    DenseX lb{B, ldb, sb_b, N, K, bvec};
    return run(la, lb, C, M, N, K, sc_m, sc_n, sc_b, batch, alpha, beta, bias, st);
  } else if (!a_k && b_k) {
    DenseX la{A, lda, sa_b, M, K, avec};
    DenseK lb{B, ldb, sb_b, N, K, bvec};
    return run(la, lb, C, M, N, K, sc_m, sc_n, sc_b, batch, alpha, beta, bias, st);
  } else {
    DenseX la{A, lda, sa_b, M, K, avec};
    DenseX lb{B, ldb, sb_b, N, K, bvec};
    return run(la, lb, C, M, N, K, sc_m, sc_n, sc_b, batch, alpha, beta, bias, st);
  }
}
