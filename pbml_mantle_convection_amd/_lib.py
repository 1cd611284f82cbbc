"""ctypes binding of libmantle_hip.so (the C ABI declared in include/mantle_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel call
fails, a RuntimeError is raised.  PyTorch is plumbing only (device memory, streams)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MANTLE_LIB", os.path.join(_HERE, "libmantle_hip.so"))

MC_F32, MC_BF16, MC_MIX16 = 0, 1, 2
PAD_MODES = {"zeros": 0, "constant": 0, "replicate": 1, "reflect": 2}
ACTS = {"none": 0, "gelu": 1, "relu": 2, "silu": 3, "tanh": 4, "selu": 5, "elu": 6}
POST_NONE, POST_ACT, POST_GN_ACT = 0, 1, 2
GSRC_NONE, GSRC_PLAIN, GSRC_PADFOLD, GSRC_PADFOLD_POOL, GSRC_PLAIN_POOL = 0, 1, 2, 3, 4
LOSS_SLOTS = 16
LOSS_TYPES = {"mae": 0, "mass": 1, "curl": 2}


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "c_in0", "c_in1", "c_out", "k", "pad", "pad_mode",
                                          "dtype", "sym_h", "c_out_split", "out_f32", "sym_v", "sym_hv")]


class GradSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("kind", C.c_int32), ("pad", C.c_int32), ("pad_mode", C.c_int32),
                ("pool", C.c_int32), ("hs", C.c_int32), ("ws", C.c_int32), ("c8_total", C.c_int32), ("cb_off", C.c_int32)]


class ConvPrologue(C.Structure):
    """mc_conv_prologue: GroupNorm affine + activation of the producer, applied by the consumer on load."""
    _fields_ = [("coef0", C.c_void_p), ("coef1", C.c_void_p), ("act0", C.c_int32), ("act1", C.c_int32)]


class ConvEpilogue(C.Structure):
    """mc_conv_epilogue: dz = dA * act'(z) + GroupNorm-backward partial sums in the input-gradient launch."""
    _fields_ = [("y", C.c_void_p), ("coef", C.c_void_p), ("act", C.c_int32), ("pad", C.c_int32), ("pad_mode", C.c_int32),
                ("hs", C.c_int32), ("ws", C.c_int32), ("partials", C.c_void_p), ("part_stride", C.c_int32),
                ("y_f16", C.c_int32)]


class LossDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("p_pred", C.c_int32),
                ("loss_type", C.c_int32), ("loss_scale", C.c_int32), ("loss_derivative", C.c_int32),
                ("l2", C.c_int32), ("lambda_mom", C.c_float), ("inv_h", C.c_float), ("ra", C.c_float),
                ("t_grad", C.c_int32)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_CD, _GS, _LD = C.POINTER(ConvDesc), C.POINTER(GradSrc), C.POINTER(LossDesc)
_CP, _CE = C.POINTER(ConvPrologue), C.POINTER(ConvEpilogue)

# name -> (restype, argtypes); must list EVERY symbol include/mantle_hip.h declares
SIGNATURES = {
    "mc_version": (C.c_int, []),
    "mc_strerror": (C.c_char_p, [C.c_int]),
    "mc_pack_nchw": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "mc_unpack_nchw": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "mc_pack_grad_nchw": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "mc_sum_hw": (C.c_int, [_vp, _i32, _i32, _f32, _vp, _vp]),
    "mc_packed_weight_bytes": (_sz, [_CD, _i32]),
    "mc_conv_bank_read_extent": (_sz, [_CD]),
    "mc_pack_weights": (C.c_int, [_CD, _vp, _i32, _vp, _vp]),
    "mc_conv_tiles": (_i32, [_CD]),
    "mc_conv_kernel_name": (C.c_char_p, [_CD]),
    "mc_conv2d": (C.c_int, [_CD, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mc_conv2d_fused": (C.c_int, [_CD, _vp, _vp, _CP, _vp, _vp, _vp, _vp, _vp, _CE, _vp]),
    "mc_conv2d_wgrad_fused": (C.c_int, [_CD, _vp, _vp, _CP, _vp, _vp, _vp]),
    "mc_gn_finalize_coef": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "mc_fold_blocks": (_i32, [_i32, _i32, _i32, _i32]),
    "mc_fold_padded_dz": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "mc_gn_bwd_apply_dz": (C.c_int, [_GS, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "mc_bicubic_fwd_act": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp,
                                     _vp]),
    "mc_wgrad_partial_bytes": (_sz, [_CD]),
    "mc_conv2d_wgrad": (C.c_int, [_CD, _vp, _vp, _vp, _vp, _vp]),
    "mc_conv2d_wgrad_finalize": (C.c_int, [_CD, _vp, _vp, _vp, _vp]),
    "mc_pack_weights_batched": (C.c_int, [_CD, _vp, _vp, _vp, _i32, _vp]),
    "mc_conv2d_wgrad_finalize_batched": (C.c_int, [_CD, _vp, _vp, _vp, _i32, _vp]),
    "mc_fold_padded": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mc_fold_padded2": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mc_gn_partials": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mc_gn_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "mc_gn_act_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp,
                                _vp, _vp]),
    "mc_gn_bwd_blocks": (_i32, [_i32, _i32]),
    "mc_gn_act_bwd_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _GS,
                                       _GS, _vp, _vp]),
    "mc_gn_act_bwd_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mc_gn_act_bwd_finalize_n": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mc_gn_act_bwd_apply": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                      _GS, _GS, _vp, _vp]),
    "mc_gn_act_fwd_small": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                                      _vp]),
    "mc_gn_act_bwd_small": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _GS, _GS, _vp, _vp, _vp]),
    "mc_gn_param_grads_batched": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mc_avgpool_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mc_rect_copy": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                               _vp]),
    "mc_concat_cb8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mc_gsrc_sum": (C.c_int, [_GS, _GS, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mc_bicubic_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "mc_bicubic_bwd": (C.c_int, [_GS, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                 _vp, _vp]),
    "mc_bicubic_bwd_walk": (C.c_int, [_GS, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "mc_bicubic_bwd_taps": (C.c_int, [_GS, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                      _vp, _vp]),
    "mc_bicubic_bwd_separable": (C.c_int, [_GS, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                           _vp, _vp, _vp]),
    "mc_curl_head_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "mc_curl_head_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _i64, _i64,
                                   _vp, _vp]),
    "mc_assemble_adtime_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp,
                                           _vp, _vp, _vp]),
    "mc_assemble_newad_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mc_ts_build_input": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "mc_ts_build_input_unet": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "mc_roll_forward_update": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mc_ts_wall_bc": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "mc_adnet_step": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp,
                                _vp]),
    "mc_loss_minmax": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mc_loss_fwd_bwd": (C.c_int, [_LD, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mc_loss_fused": (C.c_int, [_LD, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _i64, _i64, _vp, _vp]),
    "mc_loss_fused_blocks": (_i32, [_i32, _i32, _i32]),
    "mc_partial_sums_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _f32, _vp, _vp]),
    "mc_momentum_residual": (C.c_int, [_LD, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mc_momentum_adjoint": (C.c_int, [_LD, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mc_loss_finalize": (C.c_int, [_LD, _vp, _vp, _vp]),
    "mc_adam_step_flat": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
}

# entry points whose return value is a quantity, not a status code
VALUE_RETURNING = {"mc_version", "mc_strerror", "mc_conv_kernel_name", "mc_packed_weight_bytes", "mc_conv_bank_read_extent", "mc_conv_tiles",
                   "mc_wgrad_partial_bytes", "mc_gn_bwd_blocks", "mc_fold_blocks", "mc_loss_fused_blocks"}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m pbml_mantle_convection_amd.build_ext` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = ABI/header drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MantleHipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().mc_strerror(int(rc)).decode()
        raise MantleHipError(f"libmantle_hip: {what} failed with code {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    fn = getattr(load(), name)
    rc = fn(*args)
    if name not in VALUE_RETURNING:
        check(rc, name)
    return rc


def require_cuda(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on the MI355X (HIP device tensor): this framework's hot path is HIP-only and "
            "has no CPU fallback")
