// Implicit-GEMM convolution family: forward, adjoint (= ConvTranspose2d forward / conv input gradient)
// and weight gradient.  See include/fmi_hip.h for the contract and gemm_core.h for the machine mapping.
#include "gemm_core.h"
#include "conv3x3.h"
#include "conv_p3.h"
#include "convt3x3.h"

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool ep_scalar() {  // debug: FMI_EP_SCALAR = never use the 16-byte epilogue (read once)
  static const bool v = getenv("FMI_EP_SCALAR") != nullptr;
  return v;
}

static int check_desc(const fmi_conv_desc* d) {
  if (!d) return FMI_ERR_BAD_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 ||
      d->pad < 0 || d->x_cstride < d->C || d->y_cstride < d->K)
    return FMI_ERR_BAD_ARG;
  const int pe = d->pad, dl = d->dil > 1 ? d->dil : 1;
  if (d->dil < 0 || (dl > 1 && d->pad_mode)) return FMI_ERR_BAD_ARG;
  if (d->OH != (d->H + 2 * pe - dl * (d->kh - 1) - 1) / d->stride + 1 || d->OW != (d->W + 2 * pe - dl * (d->kw - 1) - 1) / d->stride + 1)
    return FMI_ERR_BAD_ARG;
  if (d->OH <= 0 || d->OW <= 0) return FMI_ERR_BAD_ARG;
  if (d->pad_mode == 1 && (d->pad >= d->H || d->pad >= d->W)) return FMI_ERR_UNSUPPORTED;
  if (d->pad_mode != 0 && d->pad_mode != 1) return FMI_ERR_UNSUPPORTED;
  if ((int64_t)d->N * d->H * d->W > 0x7fffffffLL / 2 || (int64_t)d->kh * d->kw * d->C > 0x3fffffffLL) return FMI_ERR_UNSUPPORTED;
  return FMI_OK;
}

// forward-geometry gather: anchors = output pixels of the conv
static ConvGeom fwd_geom(const fmi_conv_desc* d, const float* x, int n_eff) {
  ConvGeom g{};
  g.N = n_eff; g.IH = d->H; g.IW = d->W; g.C = d->C; g.cstride = d->x_cstride;
  g.GH = d->OH; g.GW = d->OW; g.S = d->stride;
  const int dl = d->dil > 1 ? d->dil : 1;  // dilation = the tap step of the gather (modules/drn.py)
  g.nty = d->kh; g.ntx = d->kw; g.dy0 = -d->pad; g.dx0 = -d->pad; g.ystep = dl; g.xstep = dl;
  g.kh0 = 0; g.kw0 = 0; g.khstep = 1; g.kwstep = 1; g.kw = d->kw;
  g.pad_mode = d->pad_mode;
  g.vec = (d->C % 4 == 0) && (d->x_cstride % 4 == 0) && aligned16(x);
  g.dGW = make_fastdiv(g.GW); g.dG = make_fastdiv(g.GH * g.GW); g.dC = make_fastdiv(g.C); g.dntx = make_fastdiv(g.ntx);
  g.img_bs = (int64_t)d->H * d->W * d->x_cstride;
  return g;
}

#ifndef FMI_HOST_EMU
// y[pixel][k] = bias[k] + residual[pixel][k] (either may be null): first pass of a split-reduction convolution
__global__ void __launch_bounds__(256) conv_split_init_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                             const float* __restrict__ res, int K, int cstride, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const int64_t o = (i / K) * cstride + k;
    float v = bias ? bias[k] : 0.f;
    if (res) v += res[o];
    y[o] = v;
  }
}
#endif

// Small feature maps with deep reductions (the 128-channel residual stacks at 32x32 and below, and the discriminator) give
// the implicit GEMM only a handful of output tiles: split the (tap, channel) reduction over several workgroups per tile.
static int conv_ksplit(int64_t M, int N, int K) {
#ifdef FMI_HOST_EMU
  return 1;
#else
  // keep the 128-wide tiles (operand reuse) and fill the chip by splitting the reduction instead of shrinking the tile
  const int64_t tiles = ceil_div64(M, 128) * ceil_div64(N, N <= 32 ? 32 : (N <= 64 ? 64 : 128));
  if (K < 512 || fmi_det()) return 1;  // reproducible mode: no split reduction (partial sums would meet through atomics)
  static const int ks_dbg = getenv("FMI_KS") ? atoi(getenv("FMI_KS")) : 0;  // experiment: force the split
  if (ks_dbg > 0) return ks_dbg;
  if (tiles >= 320) {
    // between one and two "waves" of workgroups (3 per CU x 256 CUs) the CUs that got 3 tiles set the time while the others idle
    // (VGG 28^2: 588 tiles = 2.3 per CU): splitting the work unit lets the dispatcher even it out.  Measured (TFLOP/s, split 1 / 2 / 3):
    // 588 tiles K 4608: 102 / 111 / 118;  588 tiles K 2304: 101 / 107 / 109;  512 tiles K 2304: 111 / 116 / 111;
    // 1024 tiles K 2304: 125 / 118 / 114;  1176 tiles K 2304: 101 / 97 / 100  ->  no split from ~800 tiles up
    if (tiles >= 800 || K < 2304) return 1;
    return tiles <= 520 ? 2 : 3;
  }
  // 192 .. 319 tiles: unsplit wins up to K = 2304 (256 tiles: 8 x 64^2 128 -> 128 92 vs 81 TFLOP/s split in two; the IR-SE50 stage
  // 16 x 32^2 256 -> 256 94.5 vs 89)
  if (tiles >= 192 && K < 4608) return 1;
  int64_t ks = ceil_div64(384, tiles);
  // at least 16 k-tiles per split; a handful of output tiles (M <= 128 rows: the pSp style heads at 4^2 .. 1^2) only stream the
  // weights, so more, shorter splits put more loads in flight: 4 k-tiles there
  const int64_t kmin = (M <= 128 && tiles <= 16) ? 64 : 256;
  if (ks > K / kmin) ks = K / kmin;
  return ks < 2 ? 1 : (int)(ks > (kmin == 64 ? 64 : 16) ? (kmin == 64 ? 64 : 16) : ks);
#endif
}

// y3_done: set by a path whose own launch wrote the requested piece image d->y3 (none does today: an 8-byte-per-lane epilogue form was
// measured at 1.7 TB/s, slower than the separate fmi_split3_f32 pass the caller runs otherwise)
static int fwd_impl(const fmi_conv_desc* d, const float* x, const float* wf, const float* bias, const float* residual, float* y, int act,
                    int batch_w, int64_t w_bstride, void* stream, bool* y3_done) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !wf || !y || batch_w < 1 || act < 0 || act > 2) return FMI_ERR_BAD_ARG;
  if (batch_w > 1 && batch_w != d->N) return FMI_ERR_BAD_ARG;
#ifndef FMI_HOST_EMU
  if (batch_w == 1 && fmi_conv2d_thin_supported(d) && aligned16(x))
    return fmi_conv2d_thin_fwd_f32(d, x, wf, bias, residual, y, act, stream);
#endif
  const int n_eff = batch_w > 1 ? 1 : d->N;
  ConvGeom g = fwd_geom(d, x, n_eff);
  ConvK la{x, g};
  ConvWX lb{wf, g, w_bstride, d->K, (d->K % 4 == 0) && aligned16(wf) && (w_bstride % 4 == 0)};
#ifndef FMI_HOST_EMU
  // bf16 piece images of the weights (fmi_weight_prepare_f32 writes them): [3][taps][C / 8][K][8]
  const uint16_t* w3 = (FMI_X6 && batch_w == 1 && d->C % 16 == 0 && aligned16(d->w3)) ? (const uint16_t*)d->w3 : nullptr;
#endif
  ConvEp ep{y, bias, residual, d->OH, d->OW, 1, 0, 0, d->OH, d->OW, d->y_cstride, act, g.dGW, g.dG,
            (int64_t)d->OH * d->OW * d->y_cstride};
  ep.vec = !ep_scalar() && d->K % 4 == 0 && d->y_cstride % 4 == 0 && aligned16(y) && aligned16(bias) && aligned16(residual);
#ifndef FMI_HOST_EMU
  const int ks = (act == 0 && batch_w == 1) ? conv_ksplit(g.Mdim(), d->K, g.Kdim()) : 1;
  // both operands as bf16 piece images (conv_p3.h): the activation pieces of x came from x's producer
  static const bool p3_off = getenv("FMI_P3_OFF") != nullptr;
  static const bool c3p3_off = getenv("FMI_C3P3_OFF") != nullptr;
  const bool p3_any = !p3_off && batch_w == 1 && d->x_cstride == d->C && p3_generic_ok(g, d->x3, w3, d->K);
  if (p3_any && !c3p3_off && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dil <= 1 &&
      conv3x3_p3_eligible(d->x3, w3, d->C, d->K, (int64_t)d->N * d->H * d->W)) {
    C3P3Args ca{(const uint16_t*)d->x3, w3, d->N, d->H, d->W, d->C, d->K, 0, make_fastdiv(d->W), make_fastdiv(d->H * d->W)};
    if (ks > 1) {
      const int64_t total = (int64_t)d->N * d->OH * d->OW * d->K;
      hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y, bias, residual, d->K,
                         d->y_cstride, total);
      ep.bias = nullptr;
      ep.res = nullptr;
      ep.act = 3;
      ep.vec = 0;
    }
    return launch_conv3x3_p3(ca, ep, g.Mdim(), ks, (hipStream_t)stream);
  }
  if (p3_any) {
    if (ks > 1) {
      const int64_t total = (int64_t)d->N * d->OH * d->OW * d->K;
      hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y, bias, residual, d->K,
                         d->y_cstride, total);
      ep.bias = nullptr;
      ep.res = nullptr;
      ep.act = 3;
      ep.vec = 0;
    }
    return launch_gemm_p3(ConvK3{(const uint16_t*)d->x3, g, make_fastdiv(g.ntaps())}, ConvWX3{w3, g, (int64_t)d->kh * d->kw * d->C * d->K, d->K}, ep, g.Mdim(), d->K, g.Kdim(), ks,
                          (hipStream_t)stream);
  }
  static const bool c3_off = getenv("FMI_C3_OFF") != nullptr;
  const bool c3 = !c3_off && !(FMI_EXP & 32) && batch_w == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode == 0 && d->dil <= 1 &&
                  conv3x3_eligible(x, wf, d->C, d->x_cstride, d->K, (int64_t)d->N * d->H * d->W);
  if (c3) {
    C3Args ca{x, wf, d->N, d->H, d->W, d->C, d->x_cstride, d->K, 0, make_fastdiv(d->W), make_fastdiv(d->H * d->W), w3};
    if (ks > 1) {
      const int64_t total = (int64_t)d->N * d->OH * d->OW * d->K;
      hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y, bias, residual, d->K,
                         d->y_cstride, total);
      ep.bias = nullptr;
      ep.res = nullptr;
      ep.act = 3;
      ep.vec = 0;
    }
    return launch_conv3x3(ca, ep, g.Mdim(), ks, (hipStream_t)stream);
  }
  if (ks > 1) {
    const int64_t total = (int64_t)d->N * d->OH * d->OW * d->K;
    hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y, bias, residual, d->K,
                       d->y_cstride, total);
    ep.bias = nullptr;
    ep.res = nullptr;
    ep.act = 3;
    ep.vec = 0;
    if (w3 && la.dma_ok()) return launch_gemm(la, ConvWX3{w3, g, (int64_t)d->kh * d->kw * d->C * d->K, d->K}, ep, g.Mdim(), d->K, g.Kdim(), 1, ks, (hipStream_t)stream);
    return launch_gemm(la, lb, ep, g.Mdim(), d->K, g.Kdim(), 1, ks, (hipStream_t)stream);
  }
  if (w3 && la.dma_ok()) return launch_gemm(la, ConvWX3{w3, g, (int64_t)d->kh * d->kw * d->C * d->K, d->K}, ep, g.Mdim(), d->K, g.Kdim(), 1, 1, (hipStream_t)stream);
#endif
  return launch_gemm(la, lb, ep, g.Mdim(), d->K, g.Kdim(), batch_w, 1, (hipStream_t)stream);
}

static int check_y3(const fmi_conv_desc* d, int out_c, int out_cs) {
  if (!d->y3) return FMI_OK;
  return (out_c % 16 == 0 && out_cs == out_c && aligned16(d->y3)) ? FMI_OK : FMI_ERR_BAD_ARG;
}

extern "C" int fmi_conv2d_fwd_f32(const fmi_conv_desc* d, const float* x, const float* wf, const float* bias,
                                  const float* residual, float* y, int act, int batch_w, int64_t w_bstride,
                                  void* stream) {
  if (!d) return FMI_ERR_BAD_ARG;
  int rc = check_y3(d, d->K, d->y_cstride);
  if (rc) return rc;
  bool y3_done = false;
  rc = fwd_impl(d, x, wf, bias, residual, y, act, batch_w, w_bstride, stream, &y3_done);
#ifndef FMI_HOST_EMU
  if (rc == FMI_OK && d->y3 && !y3_done) rc = fmi_split3_f32(y, d->y3, nullptr, (int64_t)d->N * d->OH * d->OW, d->K, 0, 0.f, stream);
#endif
  return rc;
}

static int dgrad_impl2(const fmi_conv_desc* d, const float* dy, const float* wt, const float* bias, const float* residual,
                       const float* mask, float mslope, float* dx, int batch_w, int64_t w_bstride, void* stream, bool* y3_done);
static int dgrad_impl(const fmi_conv_desc* d, const float* dy, const float* wt, const float* bias, const float* residual,
                      const float* mask, float mslope, float* dx, int batch_w, int64_t w_bstride, void* stream) {
  if (!d) return FMI_ERR_BAD_ARG;
  int rc = check_y3(d, d->C, d->x_cstride);
  if (rc) return rc;
  bool y3_done = false;
  rc = dgrad_impl2(d, dy, wt, bias, residual, mask, mslope, dx, batch_w, w_bstride, stream, &y3_done);
#ifndef FMI_HOST_EMU
  if (rc == FMI_OK && d->y3 && !y3_done) rc = fmi_split3_f32(dx, d->y3, nullptr, (int64_t)d->N * d->H * d->W, d->C, 0, 0.f, stream);
#endif
  return rc;
}
static int dgrad_impl2(const fmi_conv_desc* d, const float* dy, const float* wt, const float* bias, const float* residual,
                       const float* mask, float mslope, float* dx, int batch_w, int64_t w_bstride, void* stream, bool* y3_done) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !wt || !dx || batch_w < 1) return FMI_ERR_BAD_ARG;
  if (d->pad_mode != 0) return FMI_ERR_UNSUPPORTED;  // reflect: run on the padded extent, then fmi_reflect_pad_fold_f32
  if (batch_w > 1 && batch_w != d->N) return FMI_ERR_BAD_ARG;
  if (mask && bias) return FMI_ERR_UNSUPPORTED;  // the mask applies to the bare adjoint; a residual is added AFTER it (ConvEp: v * mask + res)
#ifndef FMI_HOST_EMU
  if (batch_w == 1 && !bias && !residual && !mask && fmi_conv2d_thin_supported(d) && aligned16(dx))
    return fmi_conv2d_thin_dgrad_f32(d, dy, wt, dx, stream);
  if (batch_w == 1 && !bias && !residual && !mask && d->C <= 4 && d->x_cstride == d->C) {  // thin INPUT: VGG16's first layer
    const int rc_thin = fmi_conv2d_thin_input_dgrad_f32(d, dy, wt, dx, stream);
    if (rc_thin != FMI_ERR_UNSUPPORTED) return rc_thin;
  }
#endif
  const int n_eff = batch_w > 1 ? 1 : d->N;
  const int s = d->stride;
  const int dl = d->dil > 1 ? d->dil : 1;
#ifndef FMI_HOST_EMU
  // bf16 piece images of the adjoint's weights: [3][taps][K / 8][C][8]
  const uint16_t* w3 = (FMI_X6 && batch_w == 1 && d->K % 16 == 0 && aligned16(d->w3)) ? (const uint16_t*)d->w3 : nullptr;
  const int64_t w3_stride = (int64_t)d->kh * d->kw * d->K * d->C;
#endif
  if (dl > 1 && s != 1) return FMI_ERR_UNSUPPORTED;  // adjoint of a dilated AND strided convolution: not needed by modules/drn.py
#ifndef FMI_HOST_EMU
  // Strided adjoints of small feature maps (the stride-2 style heads of the pSp encoder: 512 -> 512 at 16^2 .. 2^2) are a few
  // output tiles per sub-pixel phase with up to 128 reduction tiles each -- 0.25 ms of pure load latency per call.  If any phase
  // wants a split reduction, dx is initialised once and EVERY phase adds its (partial) sums atomically.
  bool strided_split = false;
  if (s > 1 && batch_w == 1) {
    for (int py = 0; py < s && !strided_split; ++py)
      for (int px = 0; px < s && !strided_split; ++px) {
        const int GH = (d->H - py + s - 1) / s, GW = (d->W - px + s - 1) / s;
        const int kh0 = (py + d->pad) % s, kw0 = (px + d->pad) % s;
        const int nty = kh0 < d->kh ? (d->kh - kh0 + s - 1) / s : 0, ntx = kw0 < d->kw ? (d->kw - kw0 + s - 1) / s : 0;
        if (GH > 0 && GW > 0 && conv_ksplit((int64_t)d->N * GH * GW, d->C, (int64_t)nty * ntx * d->K) > 1) strided_split = true;
      }
    if (strided_split) {
      const int64_t total = (int64_t)d->N * d->H * d->W * d->C;
      hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, bias, residual, d->C,
                         d->x_cstride, total);
    }
  }
#endif
#ifndef FMI_HOST_EMU
  // thin ConvTranspose2d(3, stride 2) on a large map: all four sub-pixel phases in one tap-reuse launch (convt3x3.h)
  static const bool ct3_off = getenv("FMI_CT3_OFF") != nullptr;
  if (!ct3_off && batch_w == 1 && !strided_split && !d->x3 && !(FMI_EXP & 32) && convt3x3_eligible(d, dy, w3)) {
    CT3Args ca{};
    ca.x = dy; ca.w3 = w3; ca.N = d->N; ca.H = d->OH; ca.W = d->OW; ca.Cred = d->K; ca.cs = d->y_cstride; ca.Nout = d->C;
    ca.dW = make_fastdiv(d->OW); ca.dHW = make_fastdiv(d->OH * d->OW);
    ConvEp ep{dx, bias, residual, d->OH, d->OW, 2, 0, 0, d->H, d->W, d->x_cstride, 0, ca.dW, ca.dHW, (int64_t)d->H * d->W * d->x_cstride};
    ep.vec = !ep_scalar() && d->C % 4 == 0 && d->x_cstride % 4 == 0 && aligned16(dx) && aligned16(bias) && aligned16(residual) && aligned16(mask);
    ep.mask = mask;
    ep.mslope = mslope;
    ca.ep = ep;
    return launch_convt3x3(ca, d->N * d->OH * d->OW, (hipStream_t)stream);
  }
#endif
  for (int py = 0; py < s; ++py) {
    for (int px = 0; px < s; ++px) {
      const int GH = (d->H - py + s - 1) / s, GW = (d->W - px + s - 1) / s;
      if (GH <= 0 || GW <= 0) continue;
      ConvGeom g{};
      g.N = n_eff; g.IH = d->OH; g.IW = d->OW; g.C = d->K; g.cstride = d->y_cstride;
      g.GH = GH; g.GW = GW; g.S = 1;
      g.kh0 = (py + d->pad) % s; g.kw0 = (px + d->pad) % s;
      g.nty = g.kh0 < d->kh ? (d->kh - g.kh0 + s - 1) / s : 0;
      g.ntx = g.kw0 < d->kw ? (d->kw - g.kw0 + s - 1) / s : 0;
      g.dy0 = (py + d->pad - g.kh0) / s; g.dx0 = (px + d->pad - g.kw0) / s;
      g.ystep = -dl; g.xstep = -dl; g.khstep = s; g.kwstep = s; g.kw = d->kw;  // dl > 1 only with s == 1 (checked above)
      g.pad_mode = 0;
      g.vec = (d->K % 4 == 0) && (d->y_cstride % 4 == 0) && aligned16(dy);
      g.dGW = make_fastdiv(GW); g.dG = make_fastdiv(GH * GW); g.dC = make_fastdiv(g.C);
      g.dntx = make_fastdiv(g.ntx > 0 ? g.ntx : 1);
      g.img_bs = (int64_t)d->OH * d->OW * d->y_cstride;
      if (g.nty == 0 || g.ntx == 0) { g.nty = 0; g.ntx = 1; }
      ConvK la{dy, g};
      ConvWX lb{wt, g, w_bstride, d->C, (d->C % 4 == 0) && aligned16(wt) && (w_bstride % 4 == 0)};
      ConvEp ep{dx, bias, residual, GH, GW, s, py, px, d->H, d->W, d->x_cstride, 0, g.dGW, g.dG,
                (int64_t)d->H * d->W * d->x_cstride};
      ep.vec = !ep_scalar() && d->C % 4 == 0 && d->x_cstride % 4 == 0 && aligned16(dx) && aligned16(bias) && aligned16(residual) && aligned16(mask);
      ep.mask = mask;
      ep.mslope = mslope;
#ifndef FMI_HOST_EMU
      static const bool p3_off = getenv("FMI_P3_OFF") != nullptr;
      const bool p3 = !p3_off && batch_w == 1 && d->y_cstride == d->K && g.Kdim() > 0 && p3_generic_ok(g, d->x3, w3, d->C);
      const int ks_p = strided_split ? conv_ksplit(g.Mdim(), d->C, g.Kdim()) : ((s == 1 && batch_w == 1) ? conv_ksplit(g.Mdim(), d->C, g.Kdim()) : 1);
      static const bool c3p3_off = getenv("FMI_C3P3_OFF") != nullptr;
      if (p3 && !c3p3_off && s == 1 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->dil <= 1 &&
          conv3x3_p3_eligible(d->x3, w3, d->K, d->C, (int64_t)d->N * d->H * d->W)) {
        C3P3Args ca{(const uint16_t*)d->x3, w3, d->N, d->OH, d->OW, d->K, d->C, 1, make_fastdiv(d->OW), make_fastdiv(d->OH * d->OW)};
        if (ks_p > 1) {
          const int64_t total = (int64_t)d->N * d->H * d->W * d->C;
          hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, bias, residual,
                             d->C, d->x_cstride, total);
          ep.bias = nullptr;
          ep.res = nullptr;
          ep.act = 3;
          ep.vec = 0;
        }
        rc = launch_conv3x3_p3(ca, ep, g.Mdim(), ks_p, (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
      if (p3) {
        if (!strided_split && ks_p > 1) {
          const int64_t total = (int64_t)d->N * d->H * d->W * d->C;
          hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, bias, residual,
                             d->C, d->x_cstride, total);
        }
        if (strided_split || ks_p > 1) {
          ep.bias = nullptr;
          ep.res = nullptr;
          ep.act = 3;
          ep.vec = 0;
        }
        rc = launch_gemm_p3(ConvK3{(const uint16_t*)d->x3, g, make_fastdiv(g.ntaps())}, ConvWX3{w3, g, w3_stride, d->C}, ep, g.Mdim(), d->C, g.Kdim(), ks_p, (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
      if (strided_split) {
        ep.bias = nullptr;
        ep.res = nullptr;
        ep.act = 3;
        ep.vec = 0;
        if (w3 && la.dma_ok()) rc = launch_gemm(la, ConvWX3{w3, g, w3_stride, d->C}, ep, g.Mdim(), d->C, g.Kdim(), 1, conv_ksplit(g.Mdim(), d->C, g.Kdim()), (hipStream_t)stream);
        else rc = launch_gemm(la, lb, ep, g.Mdim(), d->C, g.Kdim(), 1, conv_ksplit(g.Mdim(), d->C, g.Kdim()), (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
      const int ks = (s == 1 && batch_w == 1) ? conv_ksplit(g.Mdim(), d->C, g.Kdim()) : 1;
      static const bool c3_off = getenv("FMI_C3_OFF") != nullptr;
      const bool c3 = !c3_off && !(FMI_EXP & 32) && s == 1 && batch_w == 1 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->dil <= 1 &&
                      conv3x3_eligible(dy, wt, d->K, d->y_cstride, d->C, (int64_t)d->N * d->H * d->W);
      if (c3) {  // adjoint of a 3x3 stride-1 pad-1 convolution = the same convolution of dy with flipped taps
        C3Args ca{dy, wt, d->N, d->OH, d->OW, d->K, d->y_cstride, d->C, 1, make_fastdiv(d->OW), make_fastdiv(d->OH * d->OW), w3};
        if (ks > 1) {
          const int64_t total = (int64_t)d->N * d->H * d->W * d->C;
          hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, bias, residual,
                             d->C, d->x_cstride, total);
          ep.bias = nullptr;
          ep.res = nullptr;
          ep.act = 3;
          ep.vec = 0;
        }
        rc = launch_conv3x3(ca, ep, g.Mdim(), ks, (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
      if (ks > 1) {
        const int64_t total = (int64_t)d->N * d->H * d->W * d->C;
        hipLaunchKernelGGL(conv_split_init_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, bias, residual,
                           d->C, d->x_cstride, total);
        ep.bias = nullptr;
        ep.res = nullptr;
        ep.act = 3;
        ep.vec = 0;
        if (w3 && la.dma_ok()) rc = launch_gemm(la, ConvWX3{w3, g, w3_stride, d->C}, ep, g.Mdim(), d->C, g.Kdim(), 1, ks, (hipStream_t)stream);
        else rc = launch_gemm(la, lb, ep, g.Mdim(), d->C, g.Kdim(), 1, ks, (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
      if (w3 && la.dma_ok()) {
        rc = launch_gemm(la, ConvWX3{w3, g, w3_stride, d->C}, ep, g.Mdim(), d->C, g.Kdim(), 1, 1, (hipStream_t)stream);
        if (rc) return rc;
        continue;
      }
#endif
      rc = launch_gemm(la, lb, ep, g.Mdim(), d->C, g.Kdim(), batch_w, 1, (hipStream_t)stream);
      if (rc) return rc;
    }
  }
  return FMI_OK;
}
extern "C" int fmi_conv2d_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* bias,
                                    const float* residual, float* dx, int batch_w, int64_t w_bstride, void* stream) {
  return dgrad_impl(d, dy, wt, bias, residual, nullptr, 0.f, dx, batch_w, w_bstride, stream);
}
// y = ConvTranspose2d(x1, W1) + ConvTranspose2d(x2, W2) + bias for two 3x3 stride-2 transposed convolutions of the same geometry (the main
// path and the bypass of ResBlockDecoder, base_function.py:297-305) in ONE launch of convt3x3.h: the reduction runs over x1's channels,
// then over x2's.  d describes the first one as for fmi_conv2d_dgrad_f32 (d->K = channels of x1, d->C = output channels, d->H x d->W the
// output, d->w3 = the piece image of its adjoint pack); K2 = channels of the dense tensor x2, w3b its piece image.
extern "C" int fmi_conv_transpose2d_pair_f32(const fmi_conv_desc* d, const float* x1, const float* x2, int K2, const void* w3b,
                                             const float* bias, float* y, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x1 || !x2 || !w3b || !d->w3 || !y || K2 <= 0) return FMI_ERR_BAD_ARG;
#ifndef FMI_HOST_EMU
  const uint16_t* w3 = (FMI_X6 && d->K % 16 == 0 && aligned16(d->w3)) ? (const uint16_t*)d->w3 : nullptr;
  if (!convt3x3_eligible(d, x1, w3) || (K2 & 15) || !aligned16(x2) || !aligned16(w3b) || d->pad_mode != 0 || d->x3) return FMI_ERR_UNSUPPORTED;
  CT3Args ca{};
  ca.x = x1; ca.w3 = w3; ca.x2 = x2; ca.w3b = (const uint16_t*)w3b; ca.Cred2 = K2; ca.cs2 = K2;
  ca.N = d->N; ca.H = d->OH; ca.W = d->OW; ca.Cred = d->K; ca.cs = d->y_cstride; ca.Nout = d->C;
  ca.dW = make_fastdiv(d->OW); ca.dHW = make_fastdiv(d->OH * d->OW);
  ConvEp ep{y, bias, nullptr, d->OH, d->OW, 2, 0, 0, d->H, d->W, d->x_cstride, 0, ca.dW, ca.dHW, (int64_t)d->H * d->W * d->x_cstride};
  ep.vec = !ep_scalar() && d->C % 4 == 0 && d->x_cstride % 4 == 0 && aligned16(y) && aligned16(bias);
  ca.ep = ep;
  return launch_convt3x3(ca, d->N * d->OH * d->OW, (hipStream_t)stream);
#else
  return FMI_ERR_UNSUPPORTED;
#endif
}
extern "C" int fmi_conv2d_dgrad_masked_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* mask, float mask_slope,
                                           float* dx, void* stream) {
  if (!mask) return FMI_ERR_BAD_ARG;
  return dgrad_impl(d, dy, wt, nullptr, nullptr, mask, mask_slope, dx, 1, 0, stream);
}

extern "C" int fmi_conv2d_dgrad_masked_add_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* mask, float mask_slope,
                                               const float* gadd, float* dx, void* stream) {
  if (!mask || !gadd) return FMI_ERR_BAD_ARG;
  return dgrad_impl(d, dy, wt, nullptr, gadd, mask, mask_slope, dx, 1, 0, stream);
}

extern "C" int fmi_conv2d_wgrad_f32(const fmi_conv_desc* d, const float* x, const float* dy, float* dwf, float* dbias,
                                    int batch_w, int64_t w_bstride, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !dy || !dwf || batch_w < 1) return FMI_ERR_BAD_ARG;
  if (batch_w > 1 && batch_w != d->N) return FMI_ERR_BAD_ARG;
#ifndef FMI_HOST_EMU
  if (batch_w == 1 && fmi_conv2d_thin_supported(d) && aligned16(x)) return fmi_conv2d_thin_wgrad_f32(d, x, dy, dwf, dbias, stream);
#endif
#ifndef FMI_HOST_EMU
  // both operands as bf16 piece images (d->x3 = pieces of x, d->y3 = pieces of dy, INPUTS here): no split arithmetic, transposed LDS reads.
  // The bias gradient does not ride along on this path (the caller runs fmi_bias_grad_f32).
  static const bool wg3_off = getenv("FMI_WG3_OFF") != nullptr || getenv("FMI_P3_OFF") != nullptr;
  if (!wg3_off && FMI_X6 && batch_w == 1 && !dbias && wgrad_p3_ok(d, d->x3, d->y3))
    return launch_wgrad_p3(d, (const uint16_t*)d->x3, (const uint16_t*)d->y3, dwf, (hipStream_t)stream);
#endif
  const int n_eff = batch_w > 1 ? 1 : d->N;
  ConvGeom g = fwd_geom(d, x, n_eff);
  // the bias gradient rides along as one extra output row (needs a float4-aligned row index and shared weights)
  const bool fuse_bias = dbias && batch_w == 1 && (g.Kdim() % 4 == 0);
  if (dbias && !fuse_bias) return FMI_ERR_UNSUPPORTED;
  const int ones_row = fuse_bias ? g.Kdim() : -1;
  const int Mg = g.Kdim() + (fuse_bias ? 1 : 0), Ng = d->K, Kg = g.Mdim();
  WgradAX la{x, g, ones_row};
  DenseX lb{dy, (int64_t)d->y_cstride, (int64_t)d->OH * d->OW * d->y_cstride, Ng, Kg,
            (d->K % 4 == 0) && (d->y_cstride % 4 == 0) && aligned16(dy)};
  WgradEp ep{dwf, g, d->K, w_bstride, dbias, ones_row};
  // split the (long) pixel reduction over workgroups; partial sums meet through fp32 atomics
  const int64_t tiles = ceil_div64(Mg, 128) * ceil_div64(Ng, Ng <= 32 ? 32 : (Ng <= 64 ? 64 : 128));
  int64_t ksplit = 2048 / (tiles * batch_w);
  const int64_t kmax = Kg / 512;
  if (ksplit > kmax) ksplit = kmax;
  if (ksplit < 1) ksplit = 1;
  static const int wks_dbg = getenv("FMI_WKS") ? atoi(getenv("FMI_WKS")) : 0;  // experiment: force the pixel split of the weight gradient
  if (wks_dbg > 0) ksplit = wks_dbg;
  if (fmi_det()) ksplit = 1;  // reproducible mode: one workgroup per tile walks the whole pixel reduction
  if (ksplit * batch_w > 65535) ksplit = 65535 / batch_w;
  return launch_gemm(la, lb, ep, Mg, Ng, Kg, batch_w, (int)ksplit, (hipStream_t)stream);
}

#ifndef FMI_HOST_EMU
// ---------------------------------------------------------------------------------------------
// dbias[k] += sum_rows g[row*cstride + k]
// vector form (K % 4 == 0, K/4 divides 256): the tensor is streamed as float4, 256 threads cover 256/(K/4) rows per
// pass with fully coalesced 16-byte loads; threads owning the same 4 channels are combined through LDS.
// ---------------------------------------------------------------------------------------------
// reproducible mode runs ONE workgroup over all rows: its threads then add tens of thousands of terms each, so they carry fp64 sums (the
// default mode's many short fp32 chains are more accurate than one long one: measured on the whole-pSp fixture)
__global__ void __launch_bounds__(256) bias_grad_vec_f64_kernel(const float* __restrict__ g, int64_t rows, int K4, float* __restrict__ dbias) {
  __shared__ double part[256][4];
  const int cg = threadIdx.x % K4, rl = threadIdx.x / K4, RL = 256 / K4;
  double s[4] = {0, 0, 0, 0};
  for (int64_t r = rl; r < rows; r += RL) {
    const float4 v = reinterpret_cast<const float4*>(g)[r * K4 + cg];
    s[0] += v.x, s[1] += v.y, s[2] += v.z, s[3] += v.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[threadIdx.x][e] = s[e];
  __syncthreads();
  if (threadIdx.x < K4) {
    double t[4] = {0, 0, 0, 0};
    for (int l = 0; l < RL; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += part[l * K4 + threadIdx.x][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dbias + 4 * threadIdx.x + e, (float)t[e]);
  }
}
__global__ void __launch_bounds__(256) bias_grad_vec_kernel(const float* __restrict__ g, int64_t rows, int K4,
                                                            float* __restrict__ dbias, int64_t rows_per_block) {
  __shared__ float4 part[256];
  const int cg = threadIdx.x % K4, rl = threadIdx.x / K4, RL = 256 / K4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0 + rl;
  for (; r + 3 * RL < r1; r += 4 * RL) {  // four independent 16-byte loads in flight per thread
    const float4 v0 = reinterpret_cast<const float4*>(g)[r * K4 + cg];
    const float4 v1 = reinterpret_cast<const float4*>(g)[(r + RL) * K4 + cg];
    const float4 v2 = reinterpret_cast<const float4*>(g)[(r + 2 * RL) * K4 + cg];
    const float4 v3 = reinterpret_cast<const float4*>(g)[(r + 3 * RL) * K4 + cg];
    s.x += (v0.x + v1.x) + (v2.x + v3.x);
    s.y += (v0.y + v1.y) + (v2.y + v3.y);
    s.z += (v0.z + v1.z) + (v2.z + v3.z);
    s.w += (v0.w + v1.w) + (v2.w + v3.w);
  }
  for (; r < r1; r += RL) {
    const float4 v = reinterpret_cast<const float4*>(g)[r * K4 + cg];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < K4) {
    float4 t = part[threadIdx.x];
    for (int l = 1; l < RL; ++l) {
      const float4 v = part[l * K4 + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    atomicAdd(dbias + 4 * threadIdx.x + 0, t.x);
    atomicAdd(dbias + 4 * threadIdx.x + 1, t.y);
    atomicAdd(dbias + 4 * threadIdx.x + 2, t.z);
    atomicAdd(dbias + 4 * threadIdx.x + 3, t.w);
  }
}
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ g, int64_t rows, int K, int cstride,
                                                        float* __restrict__ dbias, int64_t rows_per_block) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int cg = 0; cg < K; cg += 64) {
    const int c = cg + tx;
    float s = 0.f;
    if (c < K)
      for (int64_t r = r0 + ty; r < r1; r += 4) s += g[r * cstride + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < K) atomicAdd(dbias + c, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
    __syncthreads();
  }
}

extern "C" int fmi_bias_grad_f32(const float* g, int64_t rows, int K, int cstride, float* dbias, void* stream) {
  if (!g || !dbias || rows <= 0 || K <= 0 || cstride < K) return FMI_ERR_BAD_ARG;
  int64_t blocks = ceil_div64(rows, 256);
  // every workgroup ends with K atomics onto the same K addresses and those serialise: per training step (13 calls, up to 1 GB each)
  // 2048 workgroups took 1.31 ms, 512: 0.73, 256: 0.58 (the 1 GB tensor streams at 5.5 TB/s either way)
  if (blocks > 256) blocks = 256;
  if (fmi_det()) blocks = 1;  // reproducible mode: one workgroup, one contribution per channel
  const int64_t rpb = ceil_div64(rows, blocks);
  blocks = ceil_div64(rows, rpb);
  const int K4 = K / 4;
  if (fmi_det() && K % 4 == 0 && cstride == K && K4 <= 256 && (K4 & (K4 - 1)) == 0 && (((uintptr_t)g) & 15) == 0)
    hipLaunchKernelGGL(bias_grad_vec_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g, rows, K4, dbias);
  else if (K % 4 == 0 && cstride == K && K4 <= 256 && (K4 & (K4 - 1)) == 0 && (((uintptr_t)g) & 15) == 0)
    hipLaunchKernelGGL(bias_grad_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, rows, K4, dbias, rpb);
  else
    hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, rows, K, cstride, dbias, rpb);
  return fmi_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Gradient of nn.ReflectionPad2d(pad): gx[y][x] = sum of gpad over every padded position mirroring (y, x)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reflect_fold_kernel(const float* __restrict__ gp, float* __restrict__ gx, int N,
                                                           int H, int W, int C, int pad, int64_t total) {
  const int HP = H + 2 * pad, WP = W + 2 * pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    // padded rows that map onto y: y+pad, and the mirrors pad-y (y in 1..pad), 2H-2-y+pad (y in H-1-pad..H-2)
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + pad;
    if (y >= 1 && y <= pad) ys[ny++] = pad - y;
    if (y <= H - 2 && y >= H - 1 - pad) ys[ny++] = 2 * H - 2 - y + pad;
    xs[nx++] = x + pad;
    if (x >= 1 && x <= pad) xs[nx++] = pad - x;
    if (x <= W - 2 && x >= W - 1 - pad) xs[nx++] = 2 * W - 2 - x + pad;
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += gp[(((int64_t)n * HP + ys[a]) * WP + xs[b]) * C + c];
    gx[i] = s;
  }
}

extern "C" int fmi_reflect_pad_fold_f32(const float* gpad, float* gx, int N, int H, int W, int C, int pad, void* stream) {
  if (!gpad || !gx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0 || pad >= H || pad >= W) return FMI_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(fmi_bw_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gpad, gx, N,
                     H, W, C, pad, total);
  return fmi_launch_status();
}
#endif  // FMI_HOST_EMU
