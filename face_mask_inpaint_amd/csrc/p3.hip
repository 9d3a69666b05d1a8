// bf16 piece images of fp32 activations (conv_p3.h): the standalone producer and its inverse.
//   x3[pixel][C / 16][piece][16] bf16, x = x0 + x1 + x2 exactly (x6.h: split3_pair)
// The convolution epilogue (ConvEp::y3) and the norm / activation kernels write the same image next to their fp32 result; this
// pass serves tensors whose producer cannot (ATen gradient accumulation, split reductions, pooling).  Pure bandwidth: 4 B read, 6 B
// written per element.
#include "common.h"
#include "x6.h"

// op: 0 = copy, 1 = leaky relu (p0 = slope).  y (may be null): the fp32 value the pieces were cut from.
template <int OP, bool WY>
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ x3, float* __restrict__ y, int64_t n8, float p0) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {  // i = (pixel, channel group, half)
    float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    if (OP == 1) {
      a.x = a.x > 0.f ? a.x : a.x * p0, a.y = a.y > 0.f ? a.y : a.y * p0, a.z = a.z > 0.f ? a.z : a.z * p0, a.w = a.w > 0.f ? a.w : a.w * p0;
      b.x = b.x > 0.f ? b.x : b.x * p0, b.y = b.y > 0.f ? b.y : b.y * p0, b.z = b.z > 0.f ? b.z : b.z * p0, b.w = b.w > 0.f ? b.w : b.w * p0;
    }
    if (WY) {
      reinterpret_cast<float4*>(y)[2 * i] = a;
      reinterpret_cast<float4*>(y)[2 * i + 1] = b;
    }
    uint32_t w[3][4];
    split3_pair(a.x, a.y, w[0][0], w[1][0], w[2][0]);
    split3_pair(a.z, a.w, w[0][1], w[1][1], w[2][1]);
    split3_pair(b.x, b.y, w[0][2], w[1][2], w[2][2]);
    split3_pair(b.z, b.w, w[0][3], w[1][3], w[2][3]);
    uint16_t* q = x3 + (i >> 1) * 48 + (i & 1) * 8;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<uint4*>(q + pc * 16) = make_uint4(w[pc][0], w[pc][1], w[pc][2], w[pc][3]);
  }
}

extern "C" int fmi_split3_f32(const float* x, void* x3, float* y, int64_t pixels, int C, int op, float p0, void* stream) {
  if (!x || !x3 || pixels <= 0 || C <= 0 || (C & 15) || op < 0 || op > 1) return FMI_ERR_BAD_ARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)x3 & 15) || ((uintptr_t)y & 15)) return FMI_ERR_BAD_ARG;
  const int64_t n8 = pixels * (C >> 3);
  const dim3 grid(fmi_bw_grid(n8, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* o = (uint16_t*)x3;
  if (op == 0 && !y) hipLaunchKernelGGL((split3_kernel<0, false>), grid, block, 0, st, x, o, y, n8, p0);
  else if (op == 0) hipLaunchKernelGGL((split3_kernel<0, true>), grid, block, 0, st, x, o, y, n8, p0);
  else if (!y) hipLaunchKernelGGL((split3_kernel<1, false>), grid, block, 0, st, x, o, y, n8, p0);
  else hipLaunchKernelGGL((split3_kernel<1, true>), grid, block, 0, st, x, o, y, n8, p0);
  return fmi_launch_status();
}

// The same pass over a gradient tensor dy also yields the bias gradient colsum[c] += sum over pixels dy[pixel][c] (the weight gradient with
// both operands as pieces no longer reads the fp32 dy, so its fused bias gradient went away and a separate pass re-read dy: 1.4 ms of
// the C2 step).  The grid stride is a multiple of C / 8, so a thread keeps ONE 8-channel chunk: eight running sums, then one LDS and one
// global atomic round per workgroup.  Not for the reproducible mode (atomics): the caller takes fmi_bias_grad_f32 there.
__global__ void __launch_bounds__(256) split3_colsum_kernel(const float* __restrict__ x, uint16_t* __restrict__ x3, float* __restrict__ colsum, int64_t n8, int C) {
  __shared__ float part[512];
  for (int c = threadIdx.x; c < C; c += 256) part[c] = 0.f;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int cpr = C >> 3;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t i = i0; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    s[0] += a.x, s[1] += a.y, s[2] += a.z, s[3] += a.w, s[4] += b.x, s[5] += b.y, s[6] += b.z, s[7] += b.w;
    uint32_t w[3][4];
    split3_pair(a.x, a.y, w[0][0], w[1][0], w[2][0]);
    split3_pair(a.z, a.w, w[0][1], w[1][1], w[2][1]);
    split3_pair(b.x, b.y, w[0][2], w[1][2], w[2][2]);
    split3_pair(b.z, b.w, w[0][3], w[1][3], w[2][3]);
    uint16_t* q = x3 + (i >> 1) * 48 + (i & 1) * 8;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<uint4*>(q + pc * 16) = make_uint4(w[pc][0], w[pc][1], w[pc][2], w[pc][3]);
  }
  __syncthreads();
  const int ch = (int)(i0 % cpr) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(&part[ch + e], s[e]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) atomicAdd(colsum + c, part[c]);
}

extern "C" int fmi_split3_colsum_f32(const float* x, void* x3, float* colsum, int64_t pixels, int C, void* stream) {
  if (!x || !x3 || !colsum || pixels <= 0 || C <= 0 || (C & 15)) return FMI_ERR_BAD_ARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)x3 & 15)) return FMI_ERR_BAD_ARG;
  if (C > 512 || 256 % (C >> 3) != 0) return FMI_ERR_UNSUPPORTED;  // a thread must stay on one channel chunk
  const int64_t n8 = pixels * (C >> 3);
  int grid = fmi_bw_grid(n8, 256);
  if (grid > 1024) grid = 1024;  // C atomics per workgroup
  hipLaunchKernelGGL(split3_colsum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)x3, colsum, n8, C);
  return fmi_launch_status();
}

// inverse (tests, debugging): y = x0 + x1 + x2, added in fp32 from the smallest piece up (exact: the sum has at most 24 significant bits)
__global__ void __launch_bounds__(256) merge3_kernel(const uint16_t* __restrict__ x3, float* __restrict__ y, int64_t n, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t pixel = i / C;
    const int c = (int)(i - pixel * C);
    const uint16_t* q = x3 + (pixel * (C >> 4) + (c >> 4)) * 48 + (c & 15);
    const float x0 = __uint_as_float((uint32_t)q[0] << 16), x1 = __uint_as_float((uint32_t)q[16] << 16), x2 = __uint_as_float((uint32_t)q[32] << 16);
    y[i] = (x2 + x1) + x0;
  }
}
extern "C" int fmi_merge3_f32(const void* x3, float* y, int64_t pixels, int C, void* stream) {
  if (!x3 || !y || pixels <= 0 || C <= 0 || (C & 15)) return FMI_ERR_BAD_ARG;
  const int64_t n = pixels * C;
  hipLaunchKernelGGL(merge3_kernel, dim3(fmi_bw_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3, y, n, C);
  return fmi_launch_status();
}
