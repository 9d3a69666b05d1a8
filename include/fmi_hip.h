/*
 * fmi_hip.h -- C ABI of libfmi_hip.so: MI355X (gfx950) kernels for the
 * reference-guided inpainting training hot path of syncdoth/face_mask_inpaint.
 *
 * Conventions
 *  - every entry point is extern "C", takes raw DEVICE pointers + sizes, an
 *    explicit hipStream_t (passed as void*), launches asynchronously and
 *    returns an fmi_status (0 = ok).  Nothing is allocated or retained.
 *  - activations are NHWC fp32 (channels contiguous) unless stated; "cstride"
 *    arguments are the distance in floats between consecutive pixels so that
 *    channel slices of a wider tensor can be read / written in place.
 *  - unsupported arguments fail loudly (FMI_ERR_*); the reference's CUDA
 *    upfirdn2d silently returns uninitialised memory for them
 *    (modules/psp/stylegan2/op/upfirdn2d_kernel.cu:172-268).
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *  - fmi_upfirdn2d_f32      <- upfirdn2d_op.upfirdn2d, op/upfirdn2d.cpp:12-23,
 *                              kernel op/upfirdn2d_kernel.cu:52-272
 *  - fmi_fused_bias_act_f32 <- fused.fused_bias_act, op/fused_bias_act.cpp:11-20,
 *                              kernel op/fused_bias_act_kernel.cu:18-99
 *  - everything else replaces stock ATen calls made by the nn.Modules on the
 *    path (F.conv2d / conv_transpose2d / bmm / softmax / instance_norm /
 *    interpolate / avg_pool / Adam ...); each prototype cites its call site.
 */
#ifndef FMI_HIP_H
#define FMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  FMI_OK = 0,
  FMI_ERR_BAD_ARG = 1,      /* null pointer, non-positive size, misaligned view */
  FMI_ERR_UNSUPPORTED = 2,  /* shape / mode outside what the kernels implement */
  FMI_ERR_LAUNCH = 3        /* hipGetLastError() != hipSuccess after the launch */
} fmi_status;

const char* fmi_status_string(int status);
int fmi_version(void);
/* Reproducible mode (also FMI_DETERMINISTIC=1 in the environment at load time).  Default off: split reductions meet through fp32 atomics,
 * whose arrival order changes the rounding from run to run.  On: every entry picks a decomposition with ONE contributing workgroup per
 * accumulated address (no split reductions, one-block reduction tails, gather-form adjoints), and fmi_attention_bwd_det_f32 replaces the
 * atomic dQ by per-key-block partial tiles added in a fixed order: two runs are bit-identical.  Slower; a checking mode.
 * fmi_set_deterministic returns the previous setting. */
int fmi_set_deterministic(int on);
int fmi_get_deterministic(void);

/* Kernel selection override of the bf16 convolution family (tests and A/B timing; initial value from FMI_BF16_TILE): 0 = by shape,
 * 1 = never the eight-wave tiles, 2 = no eight-phase kernel, 8 = the eight-phase kernel (csrc/conv_bf16_8ph.h) wherever it is legal.
 * mode < 0 only queries.  Returns the previous mode. */
int fmi_debug_bf16_tile(int mode);

/* ------------------------------------------------------------------------
 * Dense batched GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *   C[b] = alpha * A[b] . B[b] (+ bias[n]) + beta * C[b]
 * A is M x K with element strides (sa_m, sa_k), one of which must be 1;
 * B is K x N with (sb_k, sb_n), one of which must be 1; C has (sc_m, sc_n).
 * Replaces torch.bmm / @ at example_guided_att.py:17,31, base_function.py:433,437,
 * external_function.py:184,253 and F.linear at stylegan2/model.py:160-165.
 * ---------------------------------------------------------------------- */
int fmi_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K,
                 int64_t sa_m, int64_t sa_k, int64_t sb_k, int64_t sb_n, int64_t sc_m, int64_t sc_n,
                 int batch, int64_t sa_b, int64_t sb_b, int64_t sc_b,
                 float alpha, float beta, const float* bias, void* stream);

/* ------------------------------------------------------------------------
 * Convolution family as implicit GEMM (im2col gather -> LDS -> fp32 MFMA).
 * All three entry points are described by the FORWARD convolution
 *      y[N,OH,OW,K] = conv(x[N,H,W,C], w[K,C,kh,kw], stride, pad)
 *      OH = (H + 2*pad - kh)/stride + 1
 * A ConvTranspose2d (base_function.py:326-341) is the adjoint of that conv:
 * its forward is fmi_conv2d_dgrad_f32, its input-gradient fmi_conv2d_fwd_f32
 * and its weight-gradient fmi_conv2d_wgrad_f32 with x/dy exchanged.
 * Weights are consumed in two packed layouts produced by fmi_weight_prepare:
 *      wf[tap][C][K]   (tap = kh_i*kw + kw_i)  used by fwd, produced by wgrad
 *      wt[tap][K][C]                            used by dgrad
 * batch_w > 1 selects per-sample weights (ModulatedConv2d, stylegan2/model.py:
 * 241-279): sample n uses w + n*w_bstride and N must equal batch_w.
 * ---------------------------------------------------------------------- */
typedef struct {
  int N, H, W, C;    /* x: conv input (the larger image for stride > 1) */
  int OH, OW, K;     /* y: conv output */
  int x_cstride;     /* floats between consecutive pixels of x (>= C) */
  int y_cstride;     /* floats between consecutive pixels of y (>= K) */
  int kh, kw, stride, pad;
  int pad_mode;      /* 0 = zeros, 1 = reflect (nn.ReflectionPad2d, base_function.py:390) */
  int dil;           /* dilation (modules/drn.py: 2 / 4 in the dilated stages); 0 or 1 = none.  fp32 family only, stride 1 for the adjoint */
  const void* w3;    /* optional (NULL = none): the weight operand of THIS call -- wf for fwd, wt for dgrad -- once more as the three bf16
                        piece images fmi_weight_prepare_f32 writes (entry.wf3 / entry.wt3).  fp32 products run as six bf16 MFMAs on exact
                        three-way splits of both operands; with the pieces at hand only the activations are split inside the kernel.
                        Results do not depend on whether it is given. */
  const void* x3;    /* optional (NULL = none): the ACTIVATION operand of this call -- x for fwd, dy for dgrad -- once more as a bf16 piece
                        image x3[pixel][C/16][piece 0..2][16] (fmi_split3_f32, or the y3 output of the call that produced the tensor).
                        Needs a dense tensor (pixel pitch == channels), channels % 16 == 0, zero padding and w3; the convolution then
                        runs without any split arithmetic (csrc/conv_p3.h).  Results do not depend on whether it is given. */
  void* y3;          /* optional (NULL = none): OUT, the result of this call -- y for fwd, dx for dgrad -- once more as such a piece image
                        (result channels % 16 == 0, dense rows), for the convolution that will consume it (today: a fmi_split3_f32 pass
                        behind the convolution launch) */
} fmi_conv_desc;

/* ------------------------------------------------------------------------
 * GPU side of the data path (dataloader.py:76-93,169-170: PIL resize, HWC -> CHW, / 255, Normalize), csrc/preproc.hip.
 * The host decodes files and computes Pillow's O(W + H) resampling tables; the pixels are produced here in integer arithmetic and
 * equal Pillow's bit for bit.
 *   fmi_resample_u8: one pass of Pillow's 8-bit resampling, out = clip8((2^21 + sum in * kk) >> 22).  axis 0 (along x):
 *     in [N][in_h][in_w][C] -> out [N][rows][out_len][C], output row y reads input row row0 + y; axis 1 (along y):
 *     in [N][in_h][in_w][C] -> out [N][out_len][in_w][C].  bounds[out_len][2] = (first input index, count), kk[out_len][ksize] int32.
 *   fmi_gather_u8_i64: NEAREST resize through index tables, out[n][y][x] = (int64) in[n][ytab[y]][xtab[x]]  (the mask, dataloader.py:80-91)
 *   fmi_u8_lut_chw_f32: out[n][c][p] = lut256[in[n][p][c]]  (v / 255 and the optional Normalize as a 256-entry table)
 * ---------------------------------------------------------------------- */
int fmi_resample_u8(const uint8_t* in, uint8_t* out, int N, int in_h, int in_w, int C, int out_len, int axis, int row0, int rows,
                    const int32_t* bounds, const int32_t* kk, int ksize, void* stream);
int fmi_gather_u8_i64(const uint8_t* in, int64_t* out, int N, int in_h, int in_w, int out_h, int out_w, const int32_t* ytab,
                      const int32_t* xtab, void* stream);
int fmi_u8_lut_chw_f32(const uint8_t* in, const float* lut256, float* out, int N, int H, int W, int C, void* stream);

/* bf16 piece image of a dense NHWC fp32 tensor (C % 16 == 0): x3[pixel][C/16][piece][16], x = x0 + x1 + x2 exactly, x0 = rn_bf16(x),
 * x1 = rn_bf16(x - x0), x2 = x - x0 - x1.  op 0: pieces of x; op 1: pieces of lrelu(x, p0) (the LeakyReLU -> conv pairs of
 * base_function.py:207-305).  y (may be NULL): also receives the fp32 value the pieces were cut from.  fmi_merge3_f32 is the inverse. */
int fmi_split3_f32(const float* x, void* x3, float* y, int64_t pixels, int C, int op, float p0, void* stream);
int fmi_merge3_f32(const void* x3, float* y, int64_t pixels, int C, void* stream);
/* pieces of a gradient tensor dy AND its bias gradient in the same pass: colsum[c] += sum over pixels dy[pixel][c] (fp32 atomics; the
 * reproducible mode takes fmi_bias_grad_f32 instead).  C <= 512 and 256 % (C / 8) == 0, else FMI_ERR_UNSUPPORTED.  Replaces the bias
 * term of Conv2d's backward (torch.nn.Conv2d in base_function.py:207-305) when the weight gradient reads dy as pieces. */
int fmi_split3_colsum_f32(const float* x, void* x3, float* colsum, int64_t pixels, int C, void* stream);

/* y = conv(x, wf) + bias[k] + residual ; bias/residual may be NULL.
 * act: 0 none, 1 tanh, 2 relu, applied last. residual has y's layout. */
int fmi_conv2d_fwd_f32(const fmi_conv_desc* d, const float* x, const float* wf, const float* bias,
                       const float* residual, float* y, int act, int batch_w, int64_t w_bstride, void* stream);
/* dx = conv_adjoint(dy, wt) + bias[c] + residual (dx layout = x's).  With
 * pad_mode = reflect the caller passes H,W of the PADDED input (see
 * fmi_reflect_pad_fold_f32). */
/* y = ConvTranspose2d(x1, W1) + ConvTranspose2d(x2, W2) + bias: the main path and the bypass of ResBlockDecoder (base_function.py:297-305:
 * two nn.ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) whose results are added) in one launch -- the reduction runs over
 * x1's channels and continues over x2's, so the 1 GB intermediate is neither written nor re-read.  d describes the FIRST transposed
 * convolution exactly as for fmi_conv2d_dgrad_f32 (d->K = channels of x1, d->C = output channels, d->H x d->W = the output, d->w3 = piece
 * image of its adjoint pack, entry.wt3 of fmi_weight_prepare_f32); x2 is dense with K2 channels, w3b its piece image.  Thin outputs on large
 * maps only (d->C <= 64, >= 32 768 input pixels, channels % 16 == 0); FMI_ERR_UNSUPPORTED otherwise: the caller then issues the two
 * fmi_conv2d_dgrad_f32 calls (the second with the first's result as residual). */
int fmi_conv_transpose2d_pair_f32(const fmi_conv_desc* d, const float* x1, const float* x2, int K2, const void* w3b, const float* bias,
                                  float* y, void* stream);
int fmi_conv2d_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* bias,
                         const float* residual, float* dx, int batch_w, int64_t w_bstride, void* stream);
/* Adjoint of  act(x) -> conv : dx = conv_adjoint(dy, wt) * act'(x), act' = (mask > 0 ? 1 : mask_slope); mask = x or act(x), dx's layout.
 * The LeakyReLU -> conv pairs of ResBlock / ResBlockEncoderOptimized (base_function.py:207-305) and the ReLU -> conv pairs inside
 * VGG16 (loss.py:45-65): the multiplication happens in the GEMM epilogue, the separate activation-backward pass disappears.
 * Zero padding, shared weights, no bias / residual. */
int fmi_conv2d_dgrad_masked_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* mask, float mask_slope, float* dx,
                                void* stream);
/* the same with dx = conv_adjoint(dy, wt) * act'(x) + gadd: x's second consumer in a ResBlock (the 1x1 bypass convolution,
 * base_function.py:242-268) hands its gradient in here, so no separate accumulation pass runs */
int fmi_conv2d_dgrad_masked_add_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* mask, float mask_slope,
                                    const float* gadd, float* dx, void* stream);
/* dwf[tap][C][K] += sum over pixels x (gathered) * dy ; fp32 atomics, caller zeroes dwf.
 * With d->x3 (piece image of x) AND d->y3 (here an INPUT: the piece image of dy) given, dbias == NULL, C % 32 == 0, K % 16 == 0 and dense
 * tensors, the product runs on the pieces (csrc/conv_p3.h: transposed LDS reads, no split arithmetic); results agree to fp32 rounding.
 * dbias (may be NULL; needs kh*kw*C % 4 == 0 and batch_w == 1): dbias[k] += sum over pixels dy[p][k], computed by the same
 * GEMM as one extra row of ones -- no separate pass over dy; caller zeroes it. */
int fmi_conv2d_wgrad_f32(const fmi_conv_desc* d, const float* x, const float* dy, float* dwf, float* dbias,
                         int batch_w, int64_t w_bstride, void* stream);
/* Thin-output 3x3 convolution (K <= 4 output channels, stride 1, pad 1, C = 4..64 a power of two): the Output block of
 * the PICNet generator, base_function.py:367-398 (LeakyReLU -> ReflectionPad2d(1) -> conv3x3(32 -> 3) -> tanh) at 1024^2.
 * Bandwidth kernels; fmi_conv2d_fwd_f32 / _wgrad_f32 / _dgrad_f32 (zero padding) route to them by themselves.
 * fmi_conv2d_thin_dgrad_f32 also takes pad_mode = reflect on the UNPADDED extent and includes the fold of the padded gradient
 * (one pass instead of fmi_conv2d_dgrad_f32 on the padded extent + fmi_reflect_pad_fold_f32). */
int fmi_conv2d_thin_supported(const fmi_conv_desc* d);
int fmi_conv2d_thin_fwd_f32(const fmi_conv_desc* d, const float* x, const float* wf, const float* bias,
                            const float* residual, float* y, int act, void* stream);
int fmi_conv2d_thin_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, float* dx, void* stream);
int fmi_conv2d_thin_wgrad_f32(const fmi_conv_desc* d, const float* x, const float* dy, float* dwf, float* dbias, void* stream);
/* The same thin-output convolution reading lrelu(x, in_slope) -- the whole Output block LeakyReLU -> ReflectionPad2d(1) -> conv3x3 -> tanh
 * (base_function.py:386-396) in ONE pass over the activation per direction: the activation is applied while the halo window is staged
 * (forward, weight gradient), the adjoint multiplies its result by lrelu'(x).  C = 32 only (fmi_conv2d_thin_lrelu_supported). */
int fmi_conv2d_thin_lrelu_supported(const fmi_conv_desc* d);
int fmi_conv2d_thin_lrelu_fwd_f32(const fmi_conv_desc* d, const float* x, float in_slope, const float* wf, const float* bias, float* y, int act,
                                  void* stream);
int fmi_conv2d_thin_lrelu_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, const float* x, float in_slope, float* dx,
                                    void* stream);
int fmi_conv2d_thin_lrelu_wgrad_f32(const fmi_conv_desc* d, const float* x, float in_slope, const float* dy, float* dwf, float* dbias,
                                    void* stream);
/* adjoint of a thin-INPUT 3x3 convolution (C <= 4, K = 4..64 a power of two, stride 1, pad 1, zeros): dx = thin-output convolution of
 * dy with flipped taps -- the input gradient of VGG16's first layer (loss.py:45-65); fmi_conv2d_dgrad_f32 routes to it by itself */
int fmi_conv2d_thin_input_dgrad_f32(const fmi_conv_desc* d, const float* dy, const float* wt, float* dx, void* stream);
/* ------------------------------------------------------------------------
 * bf16 convolution family (fp32 accumulate) for the StyleGAN2 decoder of the pSp path in configs C3 / C5
 * (stylegan2/model.py:241-279: the grouped conv2d / conv_transpose2d of ModulatedConv2d.forward).  Activations are bf16 NHWC
 * (uint16_t = raw bf16 bits), weights are bf16 copies of the fp32 packs with the REDUCTION index contiguous:
 *   forward : wnk[K][kh*kw][C]   = fmi_pack_weight_bf16(wf[tap][C][K], T=taps, A=C, B=K)
 *   adjoint : wck[C][kh*kw][K]   = fmi_pack_weight_bf16(wt[tap][K][C], T=taps, A=K, B=C)
 * Needs the reduced channel count % 32 == 0, pixel pitches % 8 == 0, 16-byte aligned bases and zero padding
 * (fmi_conv2d_bf16_supported); anything else returns FMI_ERR_UNSUPPORTED -- there is no silent fp32 detour.
 * colscale (may be NULL): [N][out channels] fp32, multiplied into the result per (sample, channel) -- the demodulation factor.
 * The adjoint handles stride 2 by sub-pixel phases, i.e. it is also the ConvTranspose2d forward of the upsampling layers.
 * ws (may be NULL): a ZEROED fp32 buffer of ws_floats >= elements of the output tensor; when given (and colscale is NULL) small
 * feature maps with a deep reduction split it over workgroups (fp32 atomics into ws, then one conversion launch).  All sub-pixel
 * phases of a stride-2 adjoint run in ONE launch; stride > 2 is FMI_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------- */
int fmi_conv2d_bf16_supported(const fmi_conv_desc* d);
int fmi_conv2d_fwd_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* wnk, const float* colscale, uint16_t* y, float* ws,
                        int64_t ws_floats, void* stream);
/* y = lrelu(conv(x, W) * colscale[n][k] + nw[0] * noise[n][oy][ox] + bias[k], slope) * gain in one launch: ModulatedConv2d on pre-scaled
 * activations + demodulation + NoiseInjection + FusedLeakyReLU (stylegan2/model.py:241-294, op/fused_act.py:30-37).  colscale [N][K],
 * noise [N][OH][OW], bias [K] may be NULL.  FMI_ERR_UNSUPPORTED where the eight-phase kernel does not apply (needs C % 64 == 0,
 * K % 4 == 0, K > 64, y 8-byte and colscale / bias 16-byte aligned). */
int fmi_conv2d_fwd_act_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* wnk, const float* colscale, const float* noise,
                            const float* nw, const float* bias, float slope, float gain, uint16_t* y, void* stream);
int fmi_conv2d_dgrad_bf16(const fmi_conv_desc* d, const uint16_t* dy, const uint16_t* wck, const float* colscale, uint16_t* dx, float* ws,
                          int64_t ws_floats, void* stream);
/* dwf[tap][C][K] (fp32, the layout of fmi_conv2d_wgrad_f32) += sum over pixels x (gathered) * dy; fp32 atomics across the pixel
 * splits, caller zeroes dwf.  Operands reach the matrix cores through ds_read_b64_tr_b16 (the reduction index -- pixels -- is the
 * slow index of both NHWC tensors).  Needs C % 32 == 0, K % 8 == 0. */
int fmi_conv2d_wgrad_bf16(const fmi_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dwf, void* stream);
/* dst[b][t][a] = bf16(src[t][a][b]) */
int fmi_pack_weight_bf16(const float* src_tab, uint16_t* dst_bta, int T, int A, int B, void* stream);
/* bf16 NHWC activations around those convolutions (C % 8 == 0; per-sample / per-channel factors, noise, biases and all reduction
 * results are fp32).  Same semantics as the _f32 entries of the same names:
 *   scale_channels      y = x * s[n][c]                    (modulation / demodulation, model.py:244-252); _gs: gs[n][c] = sum_p g x
 *   noise_bias_act      y = lrelu(x + bias[c] + nw*noise[p], alpha) * scale   (model.py:282-294,340-346; op/fused_act.py:72-85)
 *   noise_bias_act_bwd  gx = g*scale*(y>0 ? 1 : alpha); gbias[c] = sum gx; gnw = sum gx*noise   (ONE pass; either may be NULL)
 * The reductions (_gs, _bwd, torgb_bwd) WRITE their results: each row block stores its partial sums into `ws` (ws_floats fp32, 16-byte
 * aligned, contents irrelevant; the minimum is one partial row per sample, more lets more blocks run -- 2048 rows saturate) and a
 * second small launch adds them: no fp32 atomics (thousands of them on a few hundred addresses serialise), bit-reproducible.
 *   upfirdn2d_nhwc      the decoder's Blur and its gradient only: up = down = 1, square FIR of 2..4 taps (model.py:52-68)
 *   torgb               out[n][p][o<3] = sum_c x w[o][c] s[n][c] + bias[o] + skip   (ToRGB, model.py:349-369; out / skip fp32)
 *   torgb_bwd           gx (bf16), gw[3][C], gs[N][C], gbias[3] (may be NULL); ws_floats >= 2 * N * (3 C + 8) */
int fmi_scale_channels_bf16(const uint16_t* x, const float* s, uint16_t* y, int N, int64_t P, int C, void* stream);
int fmi_scale_channels_gs_bf16(const uint16_t* g, const uint16_t* x, float* gs, float* ws, int64_t ws_floats, int N, int64_t P, int C,
                               void* stream);
int fmi_noise_bias_act_bf16(const uint16_t* x, const float* bias, const float* noise, const float* nw, uint16_t* y, int64_t pixels, int C,
                            float alpha, float scale, void* stream);
int fmi_noise_bias_act_bwd_bf16(const uint16_t* g, const uint16_t* y, const float* noise, uint16_t* gx, float* gnw, float* gbias,
                                float* ws, int64_t ws_floats, int64_t pixels, int C, float alpha, float scale, void* stream);
int fmi_upfirdn2d_nhwc_bf16(const uint16_t* in, const float* kernel, uint16_t* out, int N, int in_h, int in_w, int C, int kh, int kw,
                            int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* out = lrelu(FIR4x4(in) * colscale[n][c] + nw[0] * noise[n][oy][ox] + bias[c], slope) * gain on bf16 NHWC maps in ONE pass: the Blur of
 * an upsampling StyledConv (stylegan2/model.py:88-91, 268-271) together with the demodulation (:250-252), NoiseInjection (:282-294)
 * and FusedLeakyReLU (op/fused_act.py:30-37) that follow it.  in [N][in_h][in_w][C], out [N][in_h+pad_y0+pad_y1-3][in_w+pad_x0+pad_x1-3][C];
 * kernel: 16 fp32 taps in upfirdn2d's orientation; colscale [N][C], noise [N][OH][OW], bias [C] may each be NULL; C % 32 == 0.
 * separable != 0: the caller states that the taps are an outer product k[i][j] = ky[i] kx[j] (the Blur's kernel is one by construction,
 * model.py:36-45) with k[0][0] != 0; the kernel then filters rows and columns in turn (8 multiplies per output instead of 16). */
int fmi_blur_act_bf16(const uint16_t* in, const float* kernel, uint16_t* out, int N, int in_h, int in_w, int C, int pad_x0, int pad_x1,
                      int pad_y0, int pad_y1, const float* colscale, const float* noise, const float* nw, const float* bias, float slope,
                      float gain, int separable, void* stream);
/* Adjoint of the fused output stage y = lrelu(z * colscale[n][c] + nw[0] * noise[n][p] + bias[c], slope) * gain of fmi_blur_act_bf16 /
 * fmi_conv2d_fwd_act_bf16 in one pass over g and y ([N][P][C] bf16): t = gradient wrt z (bf16), gd [N][C] = gradient wrt colscale,
 * gbias [C], gnw [1] (gd / gbias / gnw may be NULL).  z is not needed: gpre * (z d + nw noise + bias) = g y on both branches of the
 * leaky ReLU.  ws: partial sums, >= N*3*C floats; sums: N*3*C floats (both scratch).  Replaces the autograd chain through
 * FusedLeakyReLU / NoiseInjection / the demodulation product (op/fused_act.py:40-69, model.py:250-252, 282-294). */
int fmi_styled_out_bwd_bf16(const uint16_t* g, const uint16_t* y, const float* noise, const float* colscale, const float* nw,
                            const float* bias, uint16_t* t, float* gd, float* gbias, float* gnw, float* ws, int64_t ws_floats, float* sums,
                            int N, int64_t P, int C, float slope, float gain, void* stream);
int fmi_torgb_fwd_bf16(const uint16_t* x, const float* w, const float* s, const float* bias, const float* skip, float* out, int N,
                       int64_t P, int C, void* stream);
int fmi_torgb_bwd_bf16(const uint16_t* x, const float* w, const float* s, const float* g, uint16_t* gx, float* ws, int64_t ws_floats,
                       float* gw, float* gs, float* gbias, int N, int64_t P, int C, void* stream);

/* dbias[k] = sum over rows of g[rows, cstride] (caller zeroes dbias). */
int fmi_bias_grad_f32(const float* g, int64_t rows, int K, int cstride, float* dbias, void* stream);
/* fold the gradient w.r.t. a reflection-padded tensor [N,H+2p,W+2p,C] back onto [N,H,W,C]. */
int fmi_reflect_pad_fold_f32(const float* gpad, float* gx, int N, int H, int W, int C, int pad, void* stream);

/* ------------------------------------------------------------------------
 * Weight preparation, incl. SpectralNorm (external_function.py:30-41,70-72).
 * One launch handles a whole network: entry i describes one conv weight in the
 * torch layout w[rows][C*taps] (rows = K for Conv2d, = in_channels for
 * ConvTranspose2d, which is also the conv-view K).
 * If u != NULL: `iters` times { v <- normalize(W^T u); u <- normalize(W v) }; sigma = u.(W v);
 * the packed copies hold W / sigma.  If u == NULL the packed copies hold W.
 * ---------------------------------------------------------------------- */
typedef struct {
  const float* w;   /* [rows][C*taps] */
  float* u;         /* [rows] or NULL */
  float* v;         /* [C*taps] or NULL */
  float* wf;        /* out [taps][C][rows] */
  float* wt;        /* out [taps][rows][C] (may be NULL) */
  float* sigma;     /* out [1] (may be NULL when u == NULL) */
  int rows, C, taps;
  int iters;        /* power iterations of the spectral norm (external_function.py:22,36); 0 or 1 = one, as every caller in the reference */
  void* wf3;        /* out, optional: wf as three bf16 piece images [3][taps][C/8][rows][8] (C % 8 == 0, taps <= 36), see fmi_conv_desc.w3 */
  void* wt3;        /* out, optional: wt as three bf16 piece images [3][taps][rows/8][C][8] (rows % 8 == 0, taps <= 36) */
} fmi_weight_entry;
/* entries: HOST array (pointers inside are device pointers); passed to the kernels by value, <= 32 per launch */
int fmi_weight_prepare_f32(const fmi_weight_entry* entries, int count, void* stream);

typedef struct {
  const float* w;      /* [rows][C*taps] */
  const float* u;      /* live u (see DESIGN.md "u/v rebinding") or NULL */
  const float* v;
  const float* sigma;  /* sigma of the forward call this gradient belongs to */
  const float* dwf;    /* [taps][C][rows] gradient w.r.t. the packed effective weight */
  float* dw;           /* out [rows][C*taps], overwritten */
  int rows, C, taps, pad_;
} fmi_weight_grad_entry;
/* entries: HOST array; scratch_zeroed: DEVICE float[count], zero on entry (holds sum dWeff o W per tensor) */
int fmi_weight_grad_f32(const fmi_weight_grad_entry* entries, int count, float* scratch_zeroed, void* stream);

/* ------------------------------------------------------------------------
 * Row softmax over the last dimension (base_function.py:412,430,
 * example_guided_att.py:30) and its backward, in place capable.
 * ---------------------------------------------------------------------- */
int fmi_softmax_rows_f32(const float* x, float* y, int64_t rows, int cols, void* stream);
/* ds = p * (dp - sum_j p*dp) ; ds may alias dp */
int fmi_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, void* stream);

/* Fused flash-style self-attention (queries = keys, no scaling):  O_i = softmax(q q^T) V_i, V = [V1 | V2] along channels.
 * q [N,T,D], v1 [N,T,C1], v2 [N,T,C2] or NULL, outputs o1/o2 alike, lse [N,T] = log sum_j exp(q_i.q_j) (may be NULL).
 * Supported: T % 128 == 0, D in {16,32,64}, C1 % 32 == C2 % 32 == 0, (C1+C2)/32 in {2,4,8}; else FMI_ERR_UNSUPPORTED
 * (callers fall back to the GEMM + softmax composition). */
int fmi_attention_fwd_f32(const float* q, const float* v1, const float* v2, float* o1, float* o2, float* lse,
                          int N, int T, int D, int C1, int C2, void* stream);

/* Backward of fmi_attention_fwd_f32 (P recomputed from lse): gv1/gv2 [N,T,C] overwritten, gq_zeroed [N,T,D] accumulated with
 * fp32 atomics (caller zeroes it), delta_scratch [N,T] workspace.  Supported: T % 32 == 0, D in {32,64}, (C1+C2)/32 in {4,8}. */
int fmi_attention_bwd_f32(const float* q, const float* v1, const float* v2, const float* o1, const float* o2,
                          const float* go1, const float* go2, const float* lse, float* delta_scratch,
                          float* gv1, float* gv2, float* gq_zeroed, int N, int T, int D, int C1, int C2, void* stream);

/* ------------------------------------------------------------------------
 * Bandwidth-class element-wise kernels (float4 vectorised).
 * ---------------------------------------------------------------------- */
enum {
  FMI_EW_LRELU = 0,        /* y = x > 0 ? x : p0*x            (nn.LeakyReLU / ReLU with p0 = 0) */
  FMI_EW_LRELU_BWD = 1,    /* y = a * (b > 0 ? 1 : p0)        a = grad, b = forward input */
  FMI_EW_TANH_BWD = 2,     /* y = a * (1 - b*b)               b = forward output */
  FMI_EW_ADD = 3,          /* y = a + b */
  FMI_EW_SCALE = 4,        /* y = p0 * a */
  FMI_EW_AXPY = 5,         /* y = p0 * a + b */
  FMI_EW_MUL = 6,          /* y = a * b */
  FMI_EW_RELU_BWD_OUT = 7, /* y = a * (b > 0)                 b = forward output */
  FMI_EW_SOFTPLUS = 8,     /* y = log1p(exp(a)) (threshold 20, F.softplus) */
  FMI_EW_SOFTPLUS_BWD = 9, /* y = a * sigmoid(b)              b = forward input */
  FMI_EW_SUB = 10,         /* y = a - b */
  FMI_EW_RSQRT = 11,       /* y = rsqrt(a + p0) */
  FMI_EW_RSQRT_BWD = 12,   /* y = -0.5 * a * b^3          a = grad, b = forward output */
  FMI_EW_SIGMOID = 13,     /* y = 1 / (1 + exp(-a)) */
  FMI_EW_SIGMOID_BWD = 14, /* y = a * b * (1 - b)         b = forward output */
  FMI_EW_COUNT_
};
int fmi_eltwise_f32(int op, const float* a, const float* b, float* y, int64_t n, float p0, void* stream);
/* y = a * s[0] + b (b may be NULL), s is a 1-element DEVICE tensor (Auto_Attn gamma, base_function.py:439) */
int fmi_axpy_dev_f32(const float* a, const float* s, const float* b, float* y, int64_t n, void* stream);
/* out[0] += scale * sum(a*b)  (caller zeroes out) */
int fmi_dot_f32(const float* a, const float* b, int64_t n, float scale, float* out, void* stream);

/* mask helpers (train_reference_fill.py:340, loss.py:88-95, example_guided_att.py:35) */
int fmi_mask_binarise_i64(const int64_t* mask, float* out, int64_t n, void* stream);          /* (m > 0) ? 1 : 0, bit exact */
/* y[p,c] = x[p,c] * (invert ? 1 - m[p] : m[p]) */
int fmi_mask_mul_f32(const float* x, const float* m, float* y, int64_t pixels, int C, int invert, void* stream);
/* out[p, 0:C] = (1-m[p]) * ref_att[p,:] + m[p] * ref[p,:]   (out pixel stride out_cstride) */
int fmi_guide_blend_f32(const float* ref_att, const float* ref, const float* m, float* out,
                        int64_t pixels, int C, int out_cstride, void* stream);
/* backward: g has pixel stride g_cstride; d_ref_att = (1-m) g ; d_ref = m g */
int fmi_guide_blend_bwd_f32(const float* g, const float* m, float* d_ref_att, float* d_ref,
                            int64_t pixels, int C, int g_cstride, void* stream);
/* z[p, 0:Z] = o_src[p,0:Z] + softplus(o_src[p,Z:2Z]) * eps_q ; z[p, Z:2Z] likewise from o_ref/eps_p
 * (network.py:167-168,275-307 with the normal draws injected). */
int fmi_vae_sample_f32(const float* o_src, const float* o_ref, const float* eps_q, const float* eps_p,
                       float* z, int64_t pixels, int Z, void* stream);
int fmi_vae_sample_bwd_f32(const float* gz, const float* o_src, const float* o_ref, const float* eps_q,
                           const float* eps_p, float* g_src, float* g_ref, int64_t pixels, int Z, void* stream);

/* ------------------------------------------------------------------------
 * Pooling / resampling / normalisation (NHWC).
 * ---------------------------------------------------------------------- */
/* k x k mean pooling, stride k (nn.AvgPool2d(2,2) base_function.py:233; AdaptiveAvgPool2d 1024->256 model.py:79) */
int fmi_avgpool_f32(const float* x, float* y, int N, int H, int W, int C, int k, void* stream);
int fmi_avgpool_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int k, void* stream);
/* ---- bf16 activations for the IR-SE50 body of the pSp encoder (helpers.py:56-119; SURVEY.md 8a part B: bf16 compute, fp32
 * accumulate / parameters): NHWC bf16 tensors, fp32 statistics, slopes, gates and reduction results.  C % 8 == 0 (C % 4 for the norms).
 * The BatchNorm2d kernels are the instance-norm kernels above on bf16 tensors (gadd of the backward may be NULL). ---- */
int fmi_instnorm_stats_bf16(const uint16_t* x, double* sums, float* stats, int N, int HW, int C, float eps, double* ws, int64_t ws_doubles,
                            void* stream);
int fmi_instnorm_apply_bf16(const uint16_t* x, const float* stats, const float* gamma, const float* beta, uint16_t* y, int N, int HW, int C,
                            float slope, void* stream);
int fmi_instnorm_bwd_reduce_bf16(const uint16_t* x, const uint16_t* gy, const float* stats, const float* gamma, const float* beta, double* red,
                                 int N, int HW, int C, float slope, double* ws, int64_t ws_doubles, void* stream);
int fmi_instnorm_bwd_apply_bf16(const uint16_t* x, const uint16_t* gy, const float* stats, const float* gamma, const float* beta,
                                const double* red, const uint16_t* gadd, uint16_t* gx, float* dgamma, float* dbeta, int N, int HW, int C,
                                float slope, void* stream);
/* nn.PReLU(C) (helpers.py:105) and its backward; ga[C] is WRITTEN (partials workspace ws of >= C floats, ideally 512 C) */
int fmi_prelu_bf16(const uint16_t* x, const float* a, uint16_t* y, int64_t rows, int C, void* stream);
int fmi_prelu_bwd_bf16(const uint16_t* g, const uint16_t* x, const float* a, uint16_t* gx, float* ga, float* ws, int64_t ws_floats, int64_t rows,
                       int C, void* stream);
/* SE gate x residual add (helpers.py:64-72,116-118), plain residual add, AdaptiveAvgPool2d(1) -> fp32 [N][C] (ws >= N C floats), the
 * gradient of the pooled branch joined with the scaled branch's (gx = g + gpool[n][c] / P), MaxPool2d(1, stride) and its backward */
int fmi_scale_channels_add_bf16(const uint16_t* x, const float* s, const uint16_t* res, uint16_t* y, int N, int64_t P, int C, void* stream);
int fmi_add_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, int64_t n, void* stream);
int fmi_global_avgpool_bf16(const uint16_t* x, float* pooled, float* ws, int64_t ws_floats, int N, int64_t P, int C, void* stream);
int fmi_add_bcast_bf16(const uint16_t* g, const float* gpool, uint16_t* gx, int N, int64_t P, int C, void* stream);
/* fp32 twin: gx[n][p][c] = g[n][p][c] + gpool[n][c] / P -- the two gradients of an SE module's input (helpers.py:38-54: avg_pool and the
 * gated product both read it) in one pass.  C % 4 == 0, 16-byte aligned pointers. */
int fmi_add_bcast_f32(const float* g, const float* gpool, float* gx, int N, int64_t P, int C, void* stream);
int fmi_subsample_bf16(const uint16_t* x, uint16_t* y, int N, int H, int W, int C, int stride, int backward, void* stream);
/* y = x * s[n][c] + res: SE gate and residual add of a bottleneck_IR_SE block (helpers.py:64-72,116-118) in one pass (C % 4 == 0) */
int fmi_scale_channels_add_f32(const float* x, const float* s, const float* res, float* y, int N, int64_t P, int C, void* stream);
/* Both gradients of y = x * s[n][c] (fp32 NHWC, s [N][C]) in one pass: gx = g * s, gs[n][c] = sum_p g * x -- the SE gate of the IR-SE
 * blocks (helpers.py:38-54 under autograd), the modulation / demodulation products (model.py:244-252).  C % 4 == 0, C <= 1024, 16-byte
 * aligned pointers; ws: >= N * C floats of scratch. */
int fmi_scale_channels_bwd_f32(const float* g, const float* x, const float* s, float* gx, float* gs, float* ws, int64_t ws_floats, int N,
                               int64_t P, int C, void* stream);
/* fmi_instnorm_bwd_apply_f32 with gx += gadd: the gradient of a second consumer of x (the identity shortcut of an IR block) joins in the
 * same pass */
int fmi_instnorm_bwd_apply_add_f32(const float* x, const float* gy, const float* stats, const float* gamma, const float* beta,
                                   const double* red, const float* gadd, float* gx, float* dgamma, float* dbeta, int N, int HW, int C,
                                   float slope, void* stream);
/* y[r][:] = x[r][:] / (||x[r]|| + eps), inv_norm[r] = 1 / (||x[r]|| + eps): LPIPS' normalize_activation over the channels of a
 * pixel (criteria/lpips/utils.py:6-8, eps 1e-10) and ArcFace's l2_norm of an embedding (encoders/helpers.py:15-18, eps 0) */
int fmi_l2norm_rows_f32(const float* x, float* y, float* inv_norm, int64_t rows, int C, float eps, void* stream);
int fmi_l2norm_rows_bwd_f32(const float* g, const float* y, const float* inv_norm, float* gx, int64_t rows, int C, float eps, void* stream);
/* out[0] += scale * sum_p sum_c w[c] (fx[p][c] - fy[p][c])^2: one layer of LPIPS.forward (criteria/lpips/lpips.py:33-36: squared
 * difference, 1x1 lin convolution, spatial mean, batch sum) in one pass; caller zeroes out.  Backward: gfx / gfy may be NULL */
int fmi_lpips_layer_f32(const float* fx, const float* fy, const float* w, float* out, int64_t pixels, int C, float scale, void* stream);
int fmi_lpips_layer_bwd_f32(const float* fx, const float* fy, const float* w, const float* gout, float* gfx, float* gfy, int64_t pixels,
                            int C, float scale, void* stream);
/* dst[r][dst_c0 + j] = src[r][src_c0 + j] for j < c: channel slice / concatenation of NHWC maps (model.py:106 return_zq;
 * unet_parts.py:70 and example_guided_att.py:37 torch.cat) */
int fmi_copy_channels_f32(const float* src, float* dst, int64_t rows, int src_stride, int src_c0, int dst_stride, int dst_c0, int c,
                          void* stream);
/* nn.AdaptiveAvgPool2d for any input / output size (model.py:79,111 on non-1024 decoder outputs; psp.py:33 on outputs smaller than
 * 256; id_loss.py:19 188 -> 112): window i of an axis = [floor(i L / OL), ceil((i + 1) L / OL)).  NHWC. */
int fmi_adaptive_avgpool_f32(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream);
int fmi_adaptive_avgpool_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int OH, int OW, void* stream);
/* nn.MaxPool2d(k, stride), no padding, floor mode (criteria/lpips/networks.py: AlexNet trunk k 3 stride 2); argmax[N,OH,OW,C]
 * (may be NULL) = window position of the first maximum; the backward scatters with atomics, caller zeroes gx */
int fmi_maxpool_f32(const float* x, float* y, int32_t* argmax, int N, int H, int W, int C, int k, int stride, void* stream);
int fmi_maxpool_bwd_f32(const float* gy, const int32_t* argmax, float* gx, int N, int H, int W, int C, int k, int stride, void* stream);
/* out[p] = (float) argmax_c x[p][c] (first maximum wins, NaN is the maximum): PICNet_inference.py:100-101
 * mask_detector(src, 'train').argmax(1).float(); bit exact */
int fmi_argmax_channels_f32(const float* x, float* out, int64_t pixels, int C, void* stream);
/* 2x2 max pooling (VGG16 features, loss.py:21-25); backward routes to the first maximum */
int fmi_maxpool2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
int fmi_maxpool2_bwd_f32(const float* x, const float* gy, float* gx, int N, int H, int W, int C, void* stream);
/* bilinear, align_corners=True (model.py:10-12) with the optional per-channel normalisation
 * y = (v - mean[c]) / std[c] of the VGG input (loss.py:51-52) fused in; in/out NHWC */
int fmi_resize_bilinear_f32(const float* x, float* y, int N, int H, int W, int C, int OH, int OW,
                            const float* ch_mean, const float* ch_std, void* stream);
/* gx += ... (atomic scatter; caller zeroes gx) */
int fmi_resize_bilinear_bwd_f32(const float* gy, float* gx, int N, int H, int W, int C, int OH, int OW,
                                const float* ch_std, void* stream);
/* InstanceNorm2d(affine) (base_function.py:47): stats[n][c] = {mean, rstd}; y = act((x-mean)*rstd*g + b),
 * act = LeakyReLU(slope) fused when slope != 1.
 * ws (may be NULL): fp64 partials workspace of ws_doubles >= N*C*2 doubles (contents irrelevant; ~1024*C*2 lets every CU stream).
 * With it each workgroup stores its partial sums as one row and a second launch adds the rows: sums / red are WRITTEN.  Without it
 * the workgroups add onto the caller-ZEROED sums / red with fp64 atomics (which serialise per address: slower, and not
 * bit-reproducible). */
int fmi_instnorm_stats_f32(const float* x, double* sums /*[N][C][2]*/, float* stats /*[N][C][2]*/,
                           int N, int HW, int C, float eps, double* ws, int64_t ws_doubles, void* stream);
int fmi_instnorm_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, float* y,
                           int N, int HW, int C, float slope, void* stream);
/* nn.BatchNorm2d running statistics (helpers.py:83-113 run BatchNorm2d in training mode) from stats[C][2] = (mean, rstd) of the batch:
 * running = (1 - momentum) * running + momentum * (mean | unbiased variance), num_batches_tracked[0] += 1 (may be NULL).
 * sums[C][2] (may be NULL) = the fp64 (sum, sum of squares) fmi_instnorm_stats_f32 wrote: the variance is then taken from them
 * instead of being reconstructed as 1 / rstd^2 - eps */
int fmi_batchnorm_running_update_f32(const float* stats, const double* sums, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, int C, int64_t count, float eps, float momentum, void* stream);
/* backward of y = lrelu(IN(x)): red[n][c] = {sum g', sum g'*xhat} (ws as above), then gx; dgamma/dbeta += */
int fmi_instnorm_bwd_reduce_f32(const float* x, const float* gy, const float* stats, const float* gamma,
                                const float* beta, double* red, int N, int HW, int C, float slope, double* ws, int64_t ws_doubles,
                                void* stream);
int fmi_instnorm_bwd_apply_f32(const float* x, const float* gy, const float* stats, const float* gamma,
                               const float* beta, const double* red, float* gx, float* dgamma, float* dbeta,
                               int N, int HW, int C, float slope, void* stream);

/* ------------------------------------------------------------------------
 * Reductions for the losses (loss.py:58-64,97-118; external_function.py:110-131,187-192).
 * kind: 0 = sum |a-b| , 1 = sum (a-b)^2 , 2 = sum (a-c0)^2 (b ignored)
 * out[0] += scale * reduction  (double-precision accumulation; caller zeroes out)
 * ---------------------------------------------------------------------- */
int fmi_reduce_loss_f32(int kind, const float* a, const float* b, int64_t n, float c0, float scale,
                        float* out, void* stream);
/* gradient of the above w.r.t. a: ga = gscale[0]*scale * d/da ; gscale is a DEVICE scalar (upstream grad) */
int fmi_reduce_loss_bwd_f32(int kind, const float* a, const float* b, int64_t n, float c0, float scale,
                            const float* gscale, float* ga, void* stream);


/* SSIM metric (modules/evaluations/ssim.py:18-38): NCHW planes, window = outer product of window1d[ws], zero padding;
 * out[plane / planes_per_out] += sum over the plane's pixels of the ssim map (caller zeroes out and divides). */
int fmi_ssim_f32(const float* img1, const float* img2, const float* window1d, int ws, int planes, int H, int W,
                 int planes_per_out, float* out_zeroed, void* stream);
/* SSIM / MS-SSIM as the trainers' metric package computes them (pytorch_msssim -- third-party, absent from /root/reference and offline:
 * train_reference_fill.py:207-209,252-257, train_psp.py:176-178, PICNet_inference.py:130-131; parity unpinned): Gaussian window
 * WITHOUT padding; out_plane2[plane] = (mean of the ssim map, mean of its contrast-structure factor cs) over the (H-ws+1) x (W-ws+1)
 * valid positions.  ws_part: fp64 scratch of >= planes * 64 * 2 entries (per-workgroup partial sums, added in a fixed order:
 * reproducible).  fmi_avgpool2_pad_f32: the 2 x 2 mean pool between the scales of MS-SSIM (zero padding 0 / 1 per axis, counted in
 * the divisor); y is [planes][(H + 2 ph - 2) / 2 + 1][(W + 2 pw - 2) / 2 + 1]. */
int fmi_ssim_valid_f32(const float* img1, const float* img2, const float* window1d, int ws, int planes, int H, int W, float C1, float C2,
                       float* out_plane2, double* ws_part, int64_t ws_doubles, void* stream);
int fmi_avgpool2_pad_f32(const float* x, float* y, int planes, int H, int W, int pad_h, int pad_w, void* stream);

/* ------------------------------------------------------------------------
 * Contextual loss (external_function.py:231-274), x,y NHWC features [N,P,C].
 * ---------------------------------------------------------------------- */
int fmi_cx_channel_mean_f32(const float* y, float* mu /*[C] zeroed*/, int64_t rows, int C, void* stream);
/* out[p,:] = (x[p,:]-mu)/||x[p,:]-mu|| ; inv_norm[p] saved for backward */
int fmi_cx_normalise_f32(const float* x, const float* mu, float* out, float* inv_norm, int64_t rows, int C, void* stream);
int fmi_cx_normalise_bwd_f32(const float* g, const float* xn, const float* inv_norm, float* gx, int64_t rows, int C, void* stream);
/* from cos[N][P][P] (row i = x point, col j = y point): per-row d_min, w = exp((1 - d/(dmin+1e-5))/h), row sums;
 * cxij = w/rowsum; colmax[n][j] + argmax over i; cx[n] = mean_j colmax; loss += scale * -log(cx+1e-5)/N */
int fmi_cx_rows_f32(const float* cosm, float* cxij, float* dmin, int* argmin, float* rowsum, int N, int P, float h, void* stream);
int fmi_cx_cols_f32(const float* cxij, float* colmax, int* colarg, int N, int P, void* stream);
int fmi_cx_loss_f32(const float* colmax, float* cx /*[N]*/, float* loss /*[1] zeroed*/, int N, int P, float scale, void* stream);
/* backward: writes dcos[N][P][P] (overwrites) */
int fmi_cx_bwd_f32(const float* cxij, const float* dmin, const int* argmin, const float* rowsum, const float* cosm,
                   const int* colarg, const float* cx, const float* gscale, float* dcos, int N, int P, float h, float scale,
                   void* stream);

/* ------------------------------------------------------------------------
 * Multi-tensor Adam (torch.optim.Adam defaults as used at train_reference_fill.py:309-315).
 * ---------------------------------------------------------------------- */
typedef struct {
  float* p; const float* g; float* m; float* v; int64_t n;
} fmi_adam_entry;
int fmi_adam_step_f32(const fmi_adam_entry* entries /* HOST array */, int count, int64_t max_n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, void* stream);
/* the same Adam step with the step count in device memory: step_dev[0] is incremented first, the bias corrections are computed on
 * the device from it -- a training step captured in a HIP graph (torch.cuda.graph) replays with the right corrections */
int fmi_adam_step_dev_f32(const fmi_adam_entry* entries /* HOST array */, int count, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int* step_dev, void* stream);
/* the same with a device-side guard: when guard[0] (a device scalar, normally the loss of this step) is not finite the call changes
 * nothing -- parameters, moments and the step count keep their values: train_psp.py:328-331 ("skip the step if the loss is not finite")
 * without a host read, so it survives HIP-graph capture.  guard == NULL: fmi_adam_step_dev_f32. */
int fmi_adam_step_dev_guarded_f32(const fmi_adam_entry* entries /* HOST array */, int count, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int* step_dev, const float* guard, void* stream);
/* multi-tensor Ranger step = RAdam + Lookahead + gradient centralisation (modules/psp/ranger.py:92-184, the --optimizer ranger of
 * train_psp.py:290-293).  row_mean (scratch, rows floats) non-NULL = centralise: g -= mean of its row (tensors with more than one
 * dimension, cols = numel / shape[0]); step_size / rectified from the host (the RAdam variance rectification depends on the step
 * count only); lookahead != 0 on every k-th step: slow += alpha (p - slow), p = slow */
typedef struct {
  float* p; const float* g; float* m; float* v; float* slow; float* row_mean; int64_t n; int64_t cols;
} fmi_ranger_entry;
int fmi_ranger_step_f32(const fmi_ranger_entry* entries /* HOST array */, int count, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float step_size, int rectified, float alpha, int lookahead, void* stream);

/* ------------------------------------------------------------------------
 * The reference's own native ops (modules/psp/stylegan2/op).
 * ---------------------------------------------------------------------- */
/* in [major,in_h,in_w] (minor = 1, as op/upfirdn2d.py:96 always reshapes), kernel [kh,kw], out [major,out_h,out_w];
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1.  Generic up/down/kernel sizes are supported. */
int fmi_upfirdn2d_f32(const float* in, const float* kernel, float* out, int major, int in_h, int in_w,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* bf16 storage (uint16_t bit patterns), fp32 arithmetic, round-to-nearest-even results: the reference's op is templated on the
 * tensor dtype (op/upfirdn2d_kernel.cu:149-170 AT_DISPATCH), taps included */
int fmi_upfirdn2d_bf16(const uint16_t* in, const uint16_t* kernel, uint16_t* out, int major, int in_h, int in_w, int kh, int kw,
                       int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* x [.., C, step_b...] contiguous NCHW as in the reference: bias index = (i / step_b) % size_b.
 * act: 1 linear, 3 leaky relu; grad: 0 forward, 1 first derivative w.r.t. x using ref = forward OUTPUT, 2 second (=0).
 * bias / ref may be NULL. */
int fmi_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* out, int64_t n,
                           int step_b, int size_b, int act, int grad, float alpha, float scale, void* stream);
int fmi_fused_bias_act_bf16(const uint16_t* x, const uint16_t* bias, const uint16_t* ref, uint16_t* out, int64_t n, int step_b,
                            int size_b, int act, int grad, float alpha, float scale, void* stream);
/* dbias[c] += sum_{n,hw} g[n][c][hw] for NCHW g (grad_bias of FusedLeakyReLU, op/fused_act.py:29-36); caller zeroes dbias */
int fmi_bias_grad_nchw_f32(const float* g, int N, int C, int64_t HW, float* dbias, void* stream);
/* the same sum for a bf16 cotangent (fused_leaky_relu on a bf16 tensor); dbias stays fp32 */
int fmi_bias_grad_nchw_bf16(const void* g, int N, int C, int64_t HW, float* dbias, void* stream);
/* NHWC variant used by the product's StyledConv: y = lrelu(x + bias[c] + nw[0]*noise[p]) * scale */
int fmi_noise_bias_act_f32(const float* x, const float* bias, const float* noise, const float* nw, float* y,
                           int64_t pixels, int C, float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------
 * StyleGAN2 decoder on NHWC activations (stylegan2/model.py:187-369).  ModulatedConv2d is evaluated as
 * demod[n,o] * conv(x * s[n,c], W): algebraically the reference's per-sample weight modulation (model.py:244-250).
 * ---------------------------------------------------------------------- */
int fmi_scale_channels_f32(const float* x, const float* s, float* y, int N, int64_t P, int C, void* stream);       /* y[n,p,c] = x * s[n,c] */
/* gs[n,c] = sum_p g*x: written when a partials workspace ws (ws_floats >= N*C, contents irrelevant) is given, else += by atomics onto
 * a caller-zeroed gs */
int fmi_scale_channels_gs_f32(const float* g, const float* x, float* gs, float* ws, int64_t ws_floats, int N, int64_t P, int C, void* stream);
int fmi_sqsum_last_f32(const float* x, float* out, int64_t rows, int k, void* stream);                             /* out[r] = sum_k x[r,k]^2 */
int fmi_sqsum_last_bwd_f32(const float* x, const float* g, float* gx, int64_t rows, int k, void* stream);
/* backward of fmi_noise_bias_act_f32: gx = g*scale*(y>0?1:alpha); gnw[0] += sum gx*noise[p] (noise/gnw may be NULL) */
int fmi_noise_bias_act_bwd_f32(const float* g, const float* y, const float* noise, float* gx, float* gnw,
                               int64_t pixels, int C, float alpha, float scale, void* stream);
/* upfirdn2d with minor dimension = channels: in [N,in_h,in_w,C] -> out [N,out_h,out_w,C] */
int fmi_upfirdn2d_nhwc_f32(const float* in, const float* kernel, float* out, int N, int in_h, int in_w, int C,
                           int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* pSp encoder helpers (modules/psp/encoders/helpers.py:56-119), NHWC rows x C */
int fmi_prelu_f32(const float* x, const float* a, float* y, int64_t rows, int C, void* stream);                 /* nn.PReLU(C) */
/* ws (may be NULL): fp32 partials workspace of ws_floats >= C (contents irrelevant): ga is then WRITTEN; without it ga must be zeroed
 * and receives fp32 atomics */
int fmi_prelu_bwd_f32(const float* g, const float* x, const float* a, float* gx, float* ga, float* ws, int64_t ws_floats, int64_t rows,
                      int C, void* stream);
/* MaxPool2d(1, stride) = sub-sampling; backward != 0: x is the output gradient [N,OH,OW,C], y the (fully written) input gradient */
int fmi_subsample_f32(const float* x, float* y, int N, int H, int W, int C, int stride, int backward, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FMI_HIP_H */
