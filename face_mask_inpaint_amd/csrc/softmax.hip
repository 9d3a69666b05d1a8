// Row softmax / softmax-backward over the last dimension of a [rows][cols] fp32 matrix.
// One 256-thread workgroup per row; each pass streams the row with float4 loads (a 16384-wide row is
// 64 KB and is re-read from L2).  Used on the attention score chunks, which are sized to stay inside the
// 256 MB Infinity Cache between the GEMM that writes them and the GEMMs that read them.
#include "common.h"

__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int cols, int vec) {
  __shared__ float red[4];
  const float* xr = x + (int64_t)blockIdx.x * cols;
  float* yr = y + (int64_t)blockIdx.x * cols;
  // pass 1: per-thread online (max, sum), then combined across the block
  float m = -INFINITY, s = 0.f;
  if (vec) {
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i);
      const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      if (mx > m) {
        s *= __expf(m - mx);
        m = mx;
      }
      s += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) {
      const float v = xr[i];
      if (v > m) {
        s *= __expf(m - v);
        m = v;
      }
      s += __expf(v - m);
    }
  }
  const float gm = block_max_256(m, red);
  const float part = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = block_sum_256(part, red);
  const float inv = 1.f / gs;
  // pass 2
  if (vec) {
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + i);
      float4 o;
      o.x = expf(v.x - gm) * inv;
      o.y = expf(v.y - gm) * inv;
      o.z = expf(v.z - gm) * inv;
      o.w = expf(v.w - gm) * inv;
      *reinterpret_cast<float4*>(yr + i) = o;
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) yr[i] = expf(xr[i] - gm) * inv;
  }
}

extern "C" int fmi_softmax_rows_f32(const float* x, float* y, int64_t rows, int cols, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || rows > 0x7fffffffLL) return FMI_ERR_BAD_ARG;
  const int vec = (cols % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, cols, vec);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               float* __restrict__ ds, int cols, int vec) {
  __shared__ float red[4];
  const float* pr = p + (int64_t)blockIdx.x * cols;
  const float* dr = dp + (int64_t)blockIdx.x * cols;
  float* sr = ds + (int64_t)blockIdx.x * cols;
  float acc = 0.f;
  if (vec) {
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(pr + i);
      const float4 b = *reinterpret_cast<const float4*>(dr + i);
      acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) acc += pr[i] * dr[i];
  }
  const float dot = block_sum_256(acc, red);
  if (vec) {
    for (int i = threadIdx.x * 4; i < cols; i += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(pr + i);
      const float4 b = *reinterpret_cast<const float4*>(dr + i);
      float4 o;
      o.x = a.x * (b.x - dot);
      o.y = a.y * (b.y - dot);
      o.z = a.z * (b.z - dot);
      o.w = a.w * (b.w - dot);
      *reinterpret_cast<float4*>(sr + i) = o;
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) sr[i] = pr[i] * (dr[i] - dot);
  }
}

extern "C" int fmi_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, void* stream) {
  if (!p || !dp || !ds || rows <= 0 || cols <= 0 || rows > 0x7fffffffLL) return FMI_ERR_BAD_ARG;
  const int vec = (cols % 4 == 0) && (((uintptr_t)p & 15) == 0) && (((uintptr_t)dp & 15) == 0) && (((uintptr_t)ds & 15) == 0);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, ds, cols, vec);
  return fmi_launch_status();
}
