// Shared helpers for the gfx950 kernels of libfmi_hip.so.
#pragma once
#include <stdint.h>
#include "../../include/fmi_hip.h"

#ifdef FMI_HOST_EMU
// CPU emulation build (tests/test_host_geometry.py): the operand loaders and epilogues of gemm_core.h are
// compiled with g++ and driven by a plain triple loop, so that every piece of index algebra of the implicit
// GEMM (taps, sub-pixel phases, reflect padding, packed weight rows, output mapping, fast division) is
// checked on the CPU.  Only the MFMA tile loop itself is GPU-only.
#include <math.h>
#include <string.h>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
struct float4 {
  float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline void atomicAdd(float* p, float v) { *p += v; }
typedef void* hipStream_t;
static inline int fmi_launch_status() { return FMI_OK; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool fmi_det() { return false; }          // the launch-decomposition switches have no meaning in the emulation
static inline bool fmi_blocked_acc() { return false; }

#else  // ------------------------------- device build -------------------------------
#include <hip/hip_runtime.h>
#include <stdlib.h>

#define FMI_WAVE 64

extern "C" int fmi_deterministic_flag;  // misc.hip: reproducible mode (single-contributor reductions), see fmi_set_deterministic
static inline bool fmi_det() { return fmi_deterministic_flag != 0; }
// FMI_BLOCKED_ACC=1: blocked accumulation (a second accumulator set, flushed every 512 reduction elements) also in the default mode for
// the kernels where it costs an occupancy step; the reproducible mode always takes it, the eight-wave tiles take it for free
static inline bool fmi_blocked_acc() {
  static const bool v = getenv("FMI_BLOCKED_ACC") != nullptr && getenv("FMI_BLOCKED_ACC")[0] != '0';
  return v;
}

static inline int fmi_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FMI_OK : FMI_ERR_LAUNCH;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid size for grid-stride bandwidth kernels: enough workgroups to fill 256 CUs x 8, capped
static inline int fmi_bw_grid(int64_t work_items, int block) {
  int64_t g = ceil_div64(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide reductions for 256-thread blocks (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* lds4) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}
__device__ __forceinline__ float block_max_256(float v, float* lds4) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}

// Workgroups are dealt round-robin to the 8 XCDs (each with a private L2): hand every XCD a
// contiguous chunk of the logical tile order so that neighbouring tiles share an L2.
// Bijective for any nwg (guide: cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
#endif
