/* ORACLE (test infrastructure, not product code): plain-C restatement of the reference's two native ops and
 * of the mask binarisation.  Built by oracle/Makefile into oracle/_build/liboracle_sg2.so and called through
 * ctypes from oracle/stylegan2_cpu.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Pinned by tests/golden/stylegan2_ops.pt (outputs of the reference's own upfirdn2d_native).
 *
 * upfirdn2d  follows modules/psp/stylegan2/op/upfirdn2d.py:150-184 (upfirdn2d_native): zero-insert by `up`,
 *            pad (negative pad = crop), correlate with the FLIPPED kernel, keep every `down`-th sample.
 * fused_bias_act follows op/fused_bias_act_kernel.cu:27-47: x += b[(i/step_b)%size_b]; act*10+grad switch; * scale.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_upfirdn2d(const float* in, const float* k, float* out, int major, int in_h, int in_w, int kh, int kw,
                     int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1) {
  const int ph = in_h * up_y + pad_y0 + pad_y1, pw = in_w * up_x + pad_x0 + pad_x1; /* padded, up-sampled extent */
  if (ph < kh || pw < kw) return 1;
  const int fh = ph - kh + 1, fw = pw - kw + 1;
  const int out_h = (fh + down_y - 1) / down_y, out_w = (fw + down_x - 1) / down_x; /* [::down] */
  double* buf = (double*)malloc(sizeof(double) * (size_t)ph * pw);
  if (!buf) return 2;
  for (int m = 0; m < major; ++m) {
    const float* x = in + (size_t)m * in_h * in_w;
    for (int y = 0; y < ph; ++y)
      for (int xx = 0; xx < pw; ++xx) {
        const int uy = y - pad_y0, ux = xx - pad_x0; /* coordinate in the up-sampled image */
        double v = 0.0;
        if (uy >= 0 && ux >= 0 && uy < in_h * up_y && ux < in_w * up_x && uy % up_y == 0 && ux % up_x == 0)
          v = x[(size_t)(uy / up_y) * in_w + ux / up_x];
        buf[(size_t)y * pw + xx] = v;
      }
    for (int oy = 0; oy < out_h; ++oy)
      for (int ox = 0; ox < out_w; ++ox) {
        double acc = 0.0;
        for (int a = 0; a < kh; ++a)
          for (int b = 0; b < kw; ++b)
            acc += buf[(size_t)(oy * down_y + a) * pw + ox * down_x + b] * (double)k[(kh - 1 - a) * kw + (kw - 1 - b)];
        out[((size_t)m * out_h + oy) * out_w + ox] = (float)acc;
      }
  }
  free(buf);
  return 0;
}

int oracle_fused_bias_act(const float* x, const float* b, const float* ref, float* out, int64_t n, int step_b, int size_b,
                          int act, int grad, float alpha, float scale) {
  for (int64_t i = 0; i < n; ++i) {
    float v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    const float r = ref ? ref[i] : 0.f;
    float y;
    switch (act * 10 + grad) {
      default:
      case 10: y = v; break;
      case 11: y = v; break;
      case 12: y = 0.f; break;
      case 30: y = (v > 0.f) ? v : v * alpha; break;
      case 31: y = (r > 0.f) ? v : v * alpha; break;
      case 32: y = 0.f; break;
    }
    out[i] = y * scale;
  }
  return 0;
}

/* train_reference_fill.py:340  (mask > 0).float() */
int oracle_mask_binarise(const int64_t* m, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = m[i] > 0 ? 1.f : 0.f;
  return 0;
}
