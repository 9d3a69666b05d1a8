// How long does a wave wait between two v_mfma_f32_32x32x16_bf16 that accumulate into the SAME registers?  One wave per SIMD (as in the
// attention backward), CHAINS independent accumulators used round-robin; cycles per MFMA from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/bench_tools/experiments/mfma_chain.hip -o tools/bench_tools/experiments/_mfma_chain ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int CHAINS>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(threadIdx.x + e), b[e] = (__bf16)(float)(e + 1);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 24 / CHAINS; ++j)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
void run(float* out, unsigned long long* cyc) {
  const int iters = 1000;
  k<CHAINS><<<256, 256>>>(out, cyc, iters);
  k<CHAINS><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("chains %d: %.1f memtime ticks per MFMA (100 MHz ticks x clock ratio; compare the rows)\n", CHAINS, (double)h / (iters * 24.0));
}
int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 8);
  run<1>(out, cyc);
  run<2>(out, cyc);
  run<3>(out, cyc);
  run<4>(out, cyc);
  return 0;
}
