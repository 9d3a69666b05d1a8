// probe of v_mfma_f32_4x4x1_16b_f32 operand / result layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 + 7 * l * l), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 3, 4, 5, 9, 63}) printf("lane %2d: %.0f %.0f %.0f %.0f   (A(l)*B(blk*4+r) would be %.0f.. ; A(blk*4+r)*B(l) would be %.0f %.0f..)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l + 1.0) * (1000 + 7.0 * (l / 4 * 4) * (l / 4 * 4)), (l / 4 * 4 + 1.0) * (1000 + 7.0 * l * l), (l / 4 * 4 + 2.0) * (1000 + 7.0 * l * l));
  return 0;
}
