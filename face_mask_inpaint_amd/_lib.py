"""ctypes binding of the C ABI declared in include/fmi_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no fallback:
if the library is missing the product raises -- it never computes on the CPU or through ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfmi_hip.so")

c_f = C.c_void_p  # device float*
i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    """fmi_conv_desc (include/fmi_hip.h)."""

    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "OH", "OW", "K", "x_cstride", "y_cstride", "kh", "kw", "stride", "pad", "pad_mode", "dil")] + [("w3", C.c_void_p), ("x3", C.c_void_p), ("y3", C.c_void_p)]


class WeightEntry(C.Structure):
    _fields_ = [("w", vp), ("u", vp), ("v", vp), ("wf", vp), ("wt", vp), ("sigma", vp), ("rows", i32), ("C", i32), ("taps", i32), ("iters", i32), ("wf3", vp), ("wt3", vp)]


class WeightGradEntry(C.Structure):
    _fields_ = [("w", vp), ("u", vp), ("v", vp), ("sigma", vp), ("dwf", vp), ("dw", vp), ("rows", i32), ("C", i32), ("taps", i32), ("pad_", i32)]


class AdamEntry(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64)]


class RangerEntry(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("slow", vp), ("row_mean", vp), ("n", i64), ("cols", i64)]


PD = C.POINTER(ConvDesc)

# name -> argtypes (return type is always int status unless noted)
SIGNATURES = {
    "fmi_gemm_f32": [vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, i64, i64, i64, f32, f32, vp, vp],
    "fmi_ssim_valid_f32": [vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, i64, vp],
    "fmi_avgpool2_pad_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "fmi_resample_u8": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp],
    "fmi_gather_u8_i64": [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "fmi_u8_lut_chw_f32": [vp, vp, vp, i32, i32, i32, i32, vp],
    "fmi_split3_f32": [vp, vp, vp, i64, i32, i32, f32, vp],
    "fmi_split3_colsum_f32": [vp, vp, vp, i64, i32, vp],
    "fmi_merge3_f32": [vp, vp, i64, i32, vp],
    "fmi_conv2d_fwd_f32": [PD, vp, vp, vp, vp, vp, i32, i32, i64, vp],
    "fmi_conv2d_dgrad_f32": [PD, vp, vp, vp, vp, vp, i32, i64, vp],
    "fmi_conv_transpose2d_pair_f32": [PD, vp, vp, i32, vp, vp, vp, vp],
    "fmi_conv2d_dgrad_masked_f32": [PD, vp, vp, vp, f32, vp, vp],
    "fmi_conv2d_dgrad_masked_add_f32": [PD, vp, vp, vp, f32, vp, vp, vp],
    "fmi_conv2d_wgrad_f32": [PD, vp, vp, vp, vp, i32, i64, vp],
    "fmi_conv2d_thin_fwd_f32": [PD, vp, vp, vp, vp, vp, i32, vp],
    "fmi_conv2d_thin_dgrad_f32": [PD, vp, vp, vp, vp],
    "fmi_conv2d_thin_input_dgrad_f32": [PD, vp, vp, vp, vp],
    "fmi_conv2d_thin_lrelu_fwd_f32": [PD, vp, f32, vp, vp, vp, i32, vp],
    "fmi_conv2d_thin_lrelu_dgrad_f32": [PD, vp, vp, vp, f32, vp, vp],
    "fmi_conv2d_thin_lrelu_wgrad_f32": [PD, vp, f32, vp, vp, vp, vp],
    "fmi_conv2d_thin_wgrad_f32": [PD, vp, vp, vp, vp, vp],
    "fmi_conv2d_fwd_bf16": [PD, vp, vp, vp, vp, vp, i64, vp],
    "fmi_conv2d_dgrad_bf16": [PD, vp, vp, vp, vp, vp, i64, vp],
    "fmi_conv2d_wgrad_bf16": [PD, vp, vp, vp, vp],
    "fmi_pack_weight_bf16": [vp, vp, i32, i32, i32, vp],
    "fmi_scale_channels_bf16": [vp, vp, vp, i32, i64, i32, vp],
    "fmi_scale_channels_gs_bf16": [vp, vp, vp, vp, i64, i32, i64, i32, vp],
    "fmi_noise_bias_act_bf16": [vp, vp, vp, vp, vp, i64, i32, f32, f32, vp],
    "fmi_noise_bias_act_bwd_bf16": [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, f32, f32, vp],
    "fmi_upfirdn2d_nhwc_bf16": [vp, vp, vp] + [i32] * 14 + [vp],
    "fmi_styled_out_bwd_bf16": [vp] * 11 + [i64, vp, i32, i64, i32, f32, f32, vp],
    "fmi_conv2d_fwd_act_bf16": [PD, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp],
    "fmi_blur_act_bf16": [vp, vp, vp] + [i32] * 8 + [vp, vp, vp, vp, f32, f32, i32, vp],
    "fmi_torgb_fwd_bf16": [vp, vp, vp, vp, vp, vp, i32, i64, i32, vp],
    "fmi_torgb_bwd_bf16": [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i64, i32, vp],
    "fmi_bias_grad_f32": [vp, i64, i32, i32, vp, vp],
    "fmi_reflect_pad_fold_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "fmi_weight_prepare_f32": [vp, i32, vp],
    "fmi_weight_grad_f32": [vp, i32, vp, vp],
    "fmi_softmax_rows_f32": [vp, vp, i64, i32, vp],
    "fmi_softmax_rows_bwd_f32": [vp, vp, vp, i64, i32, vp],
    "fmi_attention_fwd_f32": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "fmi_attention_bwd_f32": [vp] * 12 + [i32] * 5 + [vp],
    "fmi_eltwise_f32": [i32, vp, vp, vp, i64, f32, vp],
    "fmi_axpy_dev_f32": [vp, vp, vp, vp, i64, vp],
    "fmi_dot_f32": [vp, vp, i64, f32, vp, vp],
    "fmi_mask_binarise_i64": [vp, vp, i64, vp],
    "fmi_mask_mul_f32": [vp, vp, vp, i64, i32, i32, vp],
    "fmi_guide_blend_f32": [vp, vp, vp, vp, i64, i32, i32, vp],
    "fmi_guide_blend_bwd_f32": [vp, vp, vp, vp, i64, i32, i32, vp],
    "fmi_vae_sample_f32": [vp, vp, vp, vp, vp, i64, i32, vp],
    "fmi_vae_sample_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "fmi_avgpool_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "fmi_avgpool_bwd_f32": [vp, vp, i32, i32, i32, i32, i32, vp],
    "fmi_maxpool2_f32": [vp, vp, i32, i32, i32, i32, vp],
    "fmi_adaptive_avgpool_f32": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_adaptive_avgpool_bwd_f32": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_maxpool_f32": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_maxpool_bwd_f32": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_argmax_channels_f32": [vp, vp, i64, i32, vp],
    "fmi_instnorm_stats_bf16": [vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
    "fmi_instnorm_apply_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "fmi_instnorm_bwd_reduce_bf16": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
    "fmi_instnorm_bwd_apply_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "fmi_prelu_bf16": [vp, vp, vp, i64, i32, vp],
    "fmi_prelu_bwd_bf16": [vp, vp, vp, vp, vp, vp, i64, i64, i32, vp],
    "fmi_scale_channels_add_bf16": [vp, vp, vp, vp, i32, i64, i32, vp],
    "fmi_add_bf16": [vp, vp, vp, i64, vp],
    "fmi_global_avgpool_bf16": [vp, vp, vp, i64, i32, i64, i32, vp],
    "fmi_add_bcast_bf16": [vp, vp, vp, i32, i64, i32, vp],
    "fmi_add_bcast_f32": [vp, vp, vp, i32, i64, i32, vp],
    "fmi_scale_channels_bwd_f32": [vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, vp],
    "fmi_subsample_bf16": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_scale_channels_add_f32": [vp, vp, vp, vp, i32, i64, i32, vp],
    "fmi_instnorm_bwd_apply_add_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "fmi_copy_channels_f32": [vp, vp, i64, i32, i32, i32, i32, i32, vp],
    "fmi_l2norm_rows_f32": [vp, vp, vp, i64, i32, f32, vp],
    "fmi_l2norm_rows_bwd_f32": [vp, vp, vp, vp, i64, i32, f32, vp],
    "fmi_lpips_layer_f32": [vp, vp, vp, vp, i64, i32, f32, vp],
    "fmi_lpips_layer_bwd_f32": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "fmi_maxpool2_bwd_f32": [vp, vp, vp, i32, i32, i32, i32, vp],
    "fmi_resize_bilinear_f32": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "fmi_resize_bilinear_bwd_f32": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "fmi_instnorm_stats_f32": [vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
    "fmi_batchnorm_running_update_f32": [vp, vp, vp, vp, vp, i32, i64, f32, f32, vp],
    "fmi_instnorm_apply_f32": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "fmi_instnorm_bwd_reduce_f32": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
    "fmi_instnorm_bwd_apply_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "fmi_reduce_loss_f32": [i32, vp, vp, i64, f32, f32, vp, vp],
    "fmi_reduce_loss_bwd_f32": [i32, vp, vp, i64, f32, f32, vp, vp, vp],
    "fmi_ssim_f32": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "fmi_cx_channel_mean_f32": [vp, vp, i64, i32, vp],
    "fmi_cx_normalise_f32": [vp, vp, vp, vp, i64, i32, vp],
    "fmi_cx_normalise_bwd_f32": [vp, vp, vp, vp, i64, i32, vp],
    "fmi_cx_rows_f32": [vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "fmi_cx_cols_f32": [vp, vp, vp, i32, i32, vp],
    "fmi_cx_loss_f32": [vp, vp, vp, i32, i32, f32, vp],
    "fmi_cx_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp],
    "fmi_adam_step_f32": [vp, i32, i64, f32, f32, f32, f32, f32, i32, vp],
    "fmi_adam_step_dev_f32": [vp, i32, f32, f32, f32, f32, f32, vp, vp],
    "fmi_adam_step_dev_guarded_f32": [vp, i32, f32, f32, f32, f32, f32, vp, vp, vp],
    "fmi_ranger_step_f32": [vp, i32, f32, f32, f32, f32, f32, f32, i32, f32, i32, vp],
    "fmi_upfirdn2d_f32": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "fmi_upfirdn2d_bf16": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "fmi_fused_bias_act_f32": [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, vp],
    "fmi_fused_bias_act_bf16": [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, vp],
    "fmi_scale_channels_f32": [vp, vp, vp, i32, i64, i32, vp],
    "fmi_scale_channels_gs_f32": [vp, vp, vp, vp, i64, i32, i64, i32, vp],
    "fmi_sqsum_last_f32": [vp, vp, i64, i32, vp],
    "fmi_sqsum_last_bwd_f32": [vp, vp, vp, i64, i32, vp],
    "fmi_noise_bias_act_bwd_f32": [vp, vp, vp, vp, vp, i64, i32, f32, f32, vp],
    "fmi_upfirdn2d_nhwc_f32": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "fmi_prelu_f32": [vp, vp, vp, i64, i32, vp],
    "fmi_prelu_bwd_f32": [vp, vp, vp, vp, vp, vp, i64, i64, i32, vp],
    "fmi_subsample_f32": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "fmi_bias_grad_nchw_f32": [vp, i32, i32, i64, vp, vp],
    "fmi_bias_grad_nchw_bf16": [vp, i32, i32, i64, vp, vp],
    "fmi_noise_bias_act_f32": [vp, vp, vp, vp, vp, i64, i32, f32, f32, vp],
}

STATUS = {0: "ok", 1: "bad argument", 2: "unsupported shape/mode", 3: "kernel launch failed"}


PREDICATES = {"fmi_debug_bf16_tile": [i32], "fmi_set_deterministic": [i32], "fmi_get_deterministic": [], "fmi_conv2d_thin_supported": [PD], "fmi_conv2d_bf16_supported": [PD], "fmi_conv2d_thin_lrelu_supported": [PD]}


class FmiError(RuntimeError):
    pass


class Library:
    """Loaded shared library with typed entry points; ``strict`` requires every declared symbol."""

    def __init__(self, path: str = LIB_PATH, strict: bool = True):
        if not os.path.exists(path):
            raise FmiError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback."
            )
        self.path = path
        self.cdll = C.CDLL(path)
        self.missing = []
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = C.c_int
            setattr(self, name[4:], self._checked(name, fn))
        for name, argtypes in PREDICATES.items():  # int-valued queries (not status codes)
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = C.c_int
            setattr(self, name[4:], fn)
        if strict and self.missing:
            raise FmiError(f"{path} lacks symbols {self.missing}")

    @staticmethod
    def _checked(name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise FmiError(f"{name} failed: {STATUS.get(rc, rc)}")

        call.__name__ = name
        return call


_LIB = None


def lib() -> Library:
    global _LIB
    if _LIB is None:
        _LIB = Library(os.environ.get("FMI_LIB_PATH", LIB_PATH))  # FMI_LIB_PATH: an experimental build of the same library
    return _LIB
