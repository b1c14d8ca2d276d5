"""ctypes binding of libdesenet_hip.so (include/desenet_hip.h).  No CPU fallback: a missing library is a hard error."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdesenet_hip.so")

DSN_F32, DSN_BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2


class dsn_tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c", C.c_int32), ("ldc", C.c_int64)]


class dsn_conv_params(C.Structure):
    _fields_ = [("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
                ("act", C.c_int32), ("accumulate", C.c_int32)]


class dsn_sgd_desc(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum_buf", C.c_void_p), ("numel", C.c_int64),
                ("group", C.c_int32), ("first_chunk", C.c_int32)]


class dsn_bn_split(C.Structure):
    _fields_ = [("split_c", C.c_int32), ("_pad", C.c_int32), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class dsn_ema_desc(C.Structure):
    _fields_ = [("ema", C.c_void_p), ("model", C.c_void_p), ("numel", C.c_int64), ("first_chunk", C.c_int32),
                ("_pad", C.c_int32)]


class dsn_pack_desc(C.Structure):
    _fields_ = [("w_oihw", C.c_void_p), ("out_fwd", C.c_void_p), ("out_dgrad", C.c_void_p), ("out_dgrad_s2", C.c_void_p),
                ("co", C.c_int32),
                ("ci", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("ci_pad", C.c_int32), ("co_pad", C.c_int32)]


class dsn_lazy_seg(C.Structure):
    _fields_ = [("c0", C.c_int32), ("c1", C.c_int32), ("ch0", C.c_int32), ("p0", C.c_int32), ("acc_c", C.c_int32),
                ("act", C.c_int32), ("acc", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("scale", C.c_void_p),
                ("shift", C.c_void_p), ("count", C.c_double), ("eps", C.c_float), ("_pad", C.c_float)]


LAZY_MAXSEG = 6


class dsn_lazy_in(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("_pad", C.c_int32), ("seg", dsn_lazy_seg * LAZY_MAXSEG)]


BNRED_MAXSEG = 2


class dsn_bnred_seg(C.Structure):
    _fields_ = [("c0", C.c_int32), ("c1", C.c_int32), ("ch0", C.c_int32), ("acc_c", C.c_int32), ("act", C.c_int32),
                ("_pad", C.c_int32), ("y", C.c_void_p), ("yld", C.c_int64), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("acc", C.c_void_p)]


class dsn_bnred(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("_pad", C.c_int32), ("seg", dsn_bnred_seg * BNRED_MAXSEG)]


class dsn_copy_seg(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("copy_bytes", C.c_int64), ("total_bytes", C.c_int64)]


class dsn_bn_final(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("acc_c", C.c_int32), ("ch0", C.c_int32), ("n", C.c_int32), ("_pad", C.c_int32),
                ("count", C.c_double), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float)]


PP_MAXSTAGE = 4


class dsn_pp_stage(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("z", C.c_void_p), ("y", C.c_void_p), ("stats", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("dy", C.c_void_p), ("dx", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dw", C.c_void_p),
                ("zld", C.c_int64), ("yld", C.c_int64), ("dyld", C.c_int64), ("P", C.c_int32), ("has_bn", C.c_int32)]


class dsn_pp_args(C.Structure):
    _fields_ = [("s", dsn_pp_stage * PP_MAXSTAGE), ("nstage", C.c_int32), ("C", C.c_int32), ("Co", C.c_int32),
                ("dtype", C.c_int32), ("act", C.c_int32), ("accumulate", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float)]


TP = C.POINTER(dsn_tensor)
CP = C.POINTER(dsn_conv_params)
vp, i32, i64, f32, u64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_double

# name -> (restype, argtypes); every symbol declared in include/desenet_hip.h
PROTOTYPES = {
    "dsn_version": (i32, []),
    "dsn_last_error": (C.c_char_p, []),
    "dsn_conv2d_fwd": (i32, [TP, vp, vp, TP, TP, CP, vp]),
    "dsn_conv2d_dgrad": (i32, [TP, vp, TP, CP, vp]),
    "dsn_conv2d_dgrad_res": (i32, [TP, vp, TP, CP, TP, vp]),
    "dsn_conv2d_dgrad_s2": (i32, [TP, vp, TP, CP, vp]),
    "dsn_conv2d_dgrad_bnred": (i32, [TP, vp, TP, CP, TP, vp, vp]),
    "dsn_conv2d_dgrad_s2_bnred": (i32, [TP, vp, TP, CP, vp, vp]),
    "dsn_conv2d_stats_rows": (i32, [i64]),
    "dsn_conv2d_fwd_stats": (i32, [TP, vp, TP, CP, vp, vp, vp]),
    "dsn_wgrad_job_bytes": (i64, []),
    "dsn_conv2d_wgrad_plan": (i32, [TP, TP, vp, i32, i32, CP, vp, i64, vp]),
    "dsn_conv2d_wgrad_plan_finish": (i32, [vp, i32, vp]),
    "dsn_conv2d_wgrad_run": (i32, [vp, i32, vp, vp]),
    "dsn_conv2d_fwd_bnacc": (i32, [TP, vp, TP, CP, vp, i64, vp]),
    "dsn_bn_act_fwd_acc": (i32, [TP, vp, i64, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, TP, TP, vp, vp]),
    "dsn_bn_finalize": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp]),
    "dsn_conv2d_wgrad_workspace_bytes": (i64, [TP, TP, CP, i32]),
    "dsn_conv2d_wgrad": (i32, [TP, TP, vp, i32, i32, CP, vp, i64, vp]),
    "dsn_pack_weight_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dsn_pack_weight_dgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dsn_box_iou": (i32, [vp, i32, vp, i32, vp, vp]),
    "dsn_seg_eval_counts": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "dsn_resize_bilinear_nchw": (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "dsn_seg_argmax_nearest": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dsn_sgd_chunk": (i32, []),
    "dsn_sgd_step": (i32, [vp, i32, i32, vp, vp]),
    "dsn_ema_step": (i32, [vp, i32, i32, vp, vp]),
    "dsn_pack_tiles": (i32, [i32, i32, i32, i32]),
    "dsn_pack_weights_multi": (i32, [vp, vp, i32, i32, vp]),
    "dsn_unpack_wgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "dsn_bn_workspace_bytes": (i64, [i32]),
    "dsn_bn_stats": (i32, [TP, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, i64, vp]),
    "dsn_channel_sum": (i32, [TP, vp, i32, vp, i64, vp]),
    "dsn_bn_act_fwd": (i32, [TP, vp, vp, i32, TP, TP, vp]),
    "dsn_bn_act_bwd": (i32, [TP, TP, vp, vp, vp, vp, i32, TP, vp, vp, i32, vp, i64, vp]),
    "dsn_bn_act_bwd_reduce": (i32, [TP, TP, vp, vp, vp, vp, i32, vp, i64, vp]),
    "dsn_bn_act_bwd_reduce_into": (i32, [TP, TP, vp, vp, vp, vp, i32, vp, i32, i32, vp]),
    "dsn_bn_act_bwd_apply": (i32, [TP, TP, vp, vp, vp, vp, i32, TP, vp, vp, i32, vp, i64, f64, f32, vp, vp]),
    "dsn_act_bwd": (i32, [TP, TP, i32, TP, vp]),
    "dsn_focus_s2d": (i32, [vp, i32, i32, i32, i32, TP, vp]),
    "dsn_adaptive_avgpool_multi": (i32, [TP, vp, i32, vp, i64, vp]),
    "dsn_bilinear_ac_multi": (i32, [vp, vp, i32, vp]),
    "dsn_bilinear_ac_bwd_multi": (i32, [vp, vp, i32, i32, vp, i64, vp]),
    "dsn_letterbox_u8": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dsn_focus_s2d_u8": (i32, [vp, i32, i32, i32, i32, TP, vp]),
    "dsn_maxpool_s1": (i32, [TP, TP, vp, i32, vp]),
    "dsn_maxpool_s1_bwd": (i32, [TP, vp, TP, i32, i32, vp]),
    "dsn_upsample_nearest2x": (i32, [TP, TP, vp]),
    "dsn_upsample_nearest2x_bwd": (i32, [TP, TP, i32, vp]),
    "dsn_bilinear_ac": (i32, [TP, TP, i32, vp]),
    "dsn_bilinear_ac_bwd": (i32, [TP, i32, TP, i32, vp, i64, vp]),
    "dsn_window_reduce_workspace_bytes": (i64, [i32, i32, i32]),
    "dsn_adaptive_avgpool": (i32, [TP, TP, vp, i64, vp]),
    "dsn_adaptive_avgpool_bwd_multi": (i32, [vp, i32, TP, i32, vp]),
    "dsn_maxpool_s1_multi": (i32, [TP, vp, vp, vp, i32, vp]),
    "dsn_maxpool_s1_bwd_multi": (i32, [vp, vp, vp, i32, TP, i32, vp]),
    "dsn_adaptive_avgpool_bwd": (i32, [TP, TP, i32, vp]),
    "dsn_copy": (i32, [TP, TP, i32, vp]),
    "dsn_ffm_scale": (i32, [TP, TP, TP, vp]),
    "dsn_ffm_scale_bwd": (i32, [TP, TP, TP, TP, TP, i32, vp, i64, vp]),
    "dsn_detect_decode": (i32, [TP, vp, vp, i64, i64, i32, i32, f32, vp, vp]),
    "dsn_detect_raw_bwd": (i32, [vp, TP, i32, i32, i32, vp]),
    "dsn_detect_decode_multi": (i32, [vp, vp, i32, vp, i64, vp, i32, i32, vp, vp, vp]),
    "dsn_detect_head_fwd_supported": (i32, [i32, i32, i32, i32]),
    "dsn_detect_head_fwd_multi": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "dsn_detect_raw_bwd_multi": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, i64, vp]),
    "dsn_nms_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "dsn_nms": (i32, [vp, i32, i32, i32, f32, f32, i32, i32, u64, i32, vp, vp, vp, i64, vp]),
    "dsn_det_loss_workspace_bytes": (i64, [i32, i32, i32, i32, i64]),
    "dsn_det_loss": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, vp,
                           vp, i64, vp]),
    "dsn_det_loss_opt": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, f32,
                               vp, i32, i32, vp, vp, i64, vp]),
    "dsn_seg_ce_workspace_bytes": (i64, []),
    "dsn_seg_ce": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp]),
    "dsn_seg_ce_up_workspace_bytes": (i64, [i32, i32, i32, i32, i32]),
    "dsn_seg_ce_up": (i32, [TP, vp, i32, i32, i32, f32, vp, TP, vp, i64, vp]),
    "dsn_cast": (i32, [vp, vp, i32, i64, vp]),
    "dsn_bn_finalize_multi": (i32, [vp, i32, vp]),
    "dsn_lazy_materialize": (i32, [TP, vp, TP, vp, TP, vp]),
    "dsn_fill32": (i32, [vp, C.c_uint32, i64, vp]),
    "dsn_copy_multi": (i32, [vp, i32, vp]),
    "dsn_add_i64": (i32, [vp, i64, i64, vp]),
    "dsn_profile_enable": (i32, [i32]),
    "dsn_profile_collect": (i32, [vp, i32]),
    "dsn_profile_kernel_count": (i32, []),
    "dsn_profile_kernel_name": (C.c_char_p, [i32]),
    "dsn_profile_dump": (i64, [vp, i64]),
    "dsn_ws_mode": (i32, [i32, i32]),
    "dsn_pp_mode": (i32, [i32]),
    "dsn_pp1_mode": (i32, [i32]),
    "dsn_pp_dir": (i32, [i32]),
    "dsn_wgrad_pp_mode": (i32, [i32]),
    "dsn_pp_stages_supported": (i32, [i32, i32, i32, i32]),
    "dsn_pp_stages_fwd": (i32, [C.POINTER(dsn_pp_args), vp]),
    "dsn_pp_stages_bwd": (i32, [C.POINTER(dsn_pp_args), vp]),
}

_lib = None


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP kernels are the only compute path of desenet_amd (there is no CPU "
                "fallback). Build it with `python -m desenet_amd.build` (hipcc --offload-arch=gfx950).")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)   # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class Unsupported(RuntimeError):
    """DSN_EUNSUPPORTED (-2): the entry point does not take this shape / layout (nothing was launched).  include/desenet_hip.h
    names the plain entry point a caller falls back to."""


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dsn_last_error().decode(errors="replace")
        raise (Unsupported if rc == -2 else RuntimeError)(f"libdesenet_hip {what} failed (status {rc}): {msg}")
