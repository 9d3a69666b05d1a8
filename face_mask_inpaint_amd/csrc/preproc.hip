// GPU side of the data path (dataloader.py:76-93,169-170 in the reference: PIL resize, HWC -> CHW, / 255, Normalize).
// The host decodes files and computes the O(W + H) resampling tables exactly as Pillow does (face_mask_inpaint_amd/preprocess.py);
// every per-pixel operation runs here, in integer arithmetic, so that the pixels equal Pillow's bit for bit:
//   fmi_resample_u8       one pass of Pillow's 8-bit resampling (Resample.c: out = clip8((2^21 + sum in * k) >> 22), integer weights)
//   fmi_gather_u8_i64     NEAREST resize through index tables, widened to int64 (the `.long()` mask of dataloader.py:91)
//   fmi_u8_lut_chw_f32    uint8 HWC -> float32 CHW through a 256-entry table (v / 255 in float64, cast, optional (x - 0.5) / 0.5)
// Pure bandwidth kernels on small tensors.
#include "common.h"

// axis 0: along x.  in [N][in_h][in_w][C], out [N][rows][out_len][C], output row y reads input row row0 + y.
// axis 1: along y.  in [N][in_h][in_w][C], out [N][out_len][in_w][C].
__global__ void __launch_bounds__(256) resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int N, int in_h, int in_w, int C,
                                                          int out_len, int axis, int row0, int rows, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    int acc = 1 << 21;
    if (axis == 0) {
      const int xx = (int)(r % out_len);
      r /= out_len;
      const int y = (int)(r % rows), n = (int)(r / rows);
      const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
      const uint8_t* p = in + (((int64_t)n * in_h + row0 + y) * in_w + x0) * C + c;
      const int* k = kk + (int64_t)xx * ksize;
      for (int x = 0; x < cnt; ++x) acc += (int)p[(int64_t)x * C] * k[x];
    } else {
      const int x = (int)(r % in_w);
      r /= in_w;
      const int yy = (int)(r % out_len), n = (int)(r / out_len);
      const int y0 = bounds[2 * yy], cnt = bounds[2 * yy + 1];
      const uint8_t* p = in + (((int64_t)n * in_h + y0) * in_w + x) * C + c;
      const int* k = kk + (int64_t)yy * ksize;
      for (int y = 0; y < cnt; ++y) acc += (int)p[(int64_t)y * in_w * C] * k[y];
    }
    acc >>= 22;  // arithmetic shift: the cubic's overshoot can make the sum negative
    out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}

extern "C" int fmi_resample_u8(const uint8_t* in, uint8_t* out, int N, int in_h, int in_w, int C, int out_len, int axis, int row0, int rows,
                               const int32_t* bounds, const int32_t* kk, int ksize, void* stream) {
  if (!in || !out || !bounds || !kk || N <= 0 || in_h <= 0 || in_w <= 0 || C <= 0 || out_len <= 0 || ksize <= 0 || (axis != 0 && axis != 1))
    return FMI_ERR_BAD_ARG;
  if (axis == 0 && (row0 < 0 || rows <= 0 || row0 + rows > in_h)) return FMI_ERR_BAD_ARG;
  if (ksize > 4096) return FMI_ERR_UNSUPPORTED;  // 255 * 2^22 * (sum |k| / 2^22 <= ~1.3) stays inside int32
  const int64_t total = axis == 0 ? (int64_t)N * rows * out_len * C : (int64_t)N * out_len * in_w * C;
  hipLaunchKernelGGL(resample_u8_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, N, in_h, in_w, C, out_len, axis, row0,
                     rows, (const int*)bounds, (const int*)kk, ksize, total);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) gather_u8_i64_kernel(const uint8_t* __restrict__ in, int64_t* __restrict__ out, int in_h, int in_w, int out_h,
                                                            int out_w, const int* __restrict__ ytab, const int* __restrict__ xtab, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % out_w);
    const int64_t r = i / out_w;
    const int y = (int)(r % out_h);
    const int64_t n = r / out_h;
    out[i] = (int64_t)in[(n * in_h + ytab[y]) * in_w + xtab[x]];
  }
}

extern "C" int fmi_gather_u8_i64(const uint8_t* in, int64_t* out, int N, int in_h, int in_w, int out_h, int out_w, const int32_t* ytab,
                                 const int32_t* xtab, void* stream) {
  if (!in || !out || !ytab || !xtab || N <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * out_h * out_w;
  hipLaunchKernelGGL(gather_u8_i64_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, in_h, in_w, out_h, out_w,
                     (const int*)ytab, (const int*)xtab, total);
  return fmi_launch_status();
}

__global__ void __launch_bounds__(256) u8_lut_chw_kernel(const uint8_t* __restrict__ in, const float* __restrict__ lut, float* __restrict__ out, int C,
                                                         int64_t hw, int64_t total) {
  __shared__ float tab[256];
  tab[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {  // i indexes the OUTPUT [n][c][p]
    const int64_t p = i % hw;
    const int64_t r = i / hw;
    const int c = (int)(r % C);
    const int64_t n = r / C;
    out[i] = tab[in[(n * hw + p) * C + c]];
  }
}

extern "C" int fmi_u8_lut_chw_f32(const uint8_t* in, const float* lut256, float* out, int N, int H, int W, int C, void* stream) {
  if (!in || !lut256 || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FMI_ERR_BAD_ARG;
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * C * hw;
  hipLaunchKernelGGL(u8_lut_chw_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, lut256, out, C, hw, total);
  return fmi_launch_status();
}
