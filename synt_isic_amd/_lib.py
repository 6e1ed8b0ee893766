"""ctypes binding of libsisic_hip.so (include/sisic.h).

This is the reference-side binding a SYNT_ISIC maintainer would add:
plain pointers and sizes, no torch types cross the boundary.  There is no CPU
fallback: if the library has not been built, ``load()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_PATH = os.environ.get("SISIC_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                             "libsisic_hip.so")
_lib: Optional[C.CDLL] = None

ABI_VERSION = 3          # include/sisic.h SISIC_ABI_VERSION

SISIC_OK = 0
SISIC_EINVAL = -1
SISIC_EHIP = -2
SISIC_ESTATE = -3
SISIC_ECANCEL = -4

c_float_p = C.POINTER(C.c_float)
c_int64_p = C.POINTER(C.c_int64)


class ConvArgs(C.Structure):
    """struct sisic_conv_args"""
    _fields_ = [
        ("in0", C.c_void_p), ("in1", C.c_void_p),
        ("c0", C.c_int), ("c1", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int),
        ("upsample", C.c_int), ("ksize", C.c_int), ("stride", C.c_int),
        ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("Cout", C.c_int),
        ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("gn_silu", C.c_int),
        ("chan_bias", C.c_void_p), ("chan_bias_stride", C.c_int),
        ("residual", C.c_void_p), ("relu", C.c_int),
        ("out", C.c_void_p), ("tile_cfg", C.c_int),
        ("w_winograd", C.c_void_p),
        ("stats_out", C.c_void_p),
        ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p), ("fin_groups", C.c_int), ("fin_eps", C.c_float),
        ("fin_scale", C.c_void_p), ("fin_shift", C.c_void_p), ("fin_mean_rstd", C.c_void_p),
    ]


class UNetConfigC(C.Structure):
    """struct sisic_unet_config"""
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("layers_per_block", C.c_int), ("n_blocks", C.c_int),
        ("block_out_channels", C.c_int * 8),
        ("down_attn", C.c_int * 8), ("up_attn", C.c_int * 8),
        ("norm_groups", C.c_int), ("norm_eps", C.c_float), ("head_dim", C.c_int),
        ("n_freqs", C.c_int), ("freqs", c_float_p),
    ]


# name -> (restype, argtypes); every symbol include/sisic.h declares
SIGNATURES = {
    "sisic_abi_version": (C.c_int, []),
    "sisic_last_error": (C.c_char_p, []),
    "sisic_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "sisic_destroy": (C.c_int, [C.c_void_p]),
    "sisic_conv_packed_numel": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "sisic_conv_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sisic_conv_winograd_numel": (C.c_int64, [C.c_int, C.c_int]),
    "sisic_conv_winograd_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sisic_conv2d": (C.c_int, [C.c_void_p, C.POINTER(ConvArgs), C.c_void_p]),
    "sisic_conv_stats_slots": (C.c_int, [C.POINTER(ConvArgs)]),
    "sisic_conv_finalizes": (C.c_int, [C.POINTER(ConvArgs)]),
    "sisic_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "sisic_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "sisic_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sisic_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "sisic_denorm_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sisic_denorm_u8_form": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "sisic_unet_create": (C.c_int, [C.c_void_p, C.POINTER(UNetConfigC), C.POINTER(C.c_void_p)]),
    "sisic_unet_destroy": (C.c_int, [C.c_void_p]),
    "sisic_unet_num_tensors": (C.c_int, [C.c_void_p]),
    "sisic_unet_tensor_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "sisic_unet_load": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), c_int64_p]),
    "sisic_unet_set_latency_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "sisic_unet_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "sisic_unet_graph_builds": (C.c_int64, [C.c_void_p]),
    "sisic_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, c_int64_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "sisic_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_int64_p, c_float_p,
                               C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.c_void_p]),
    "sisic_sample_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_int64_p, c_float_p,
                                      C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.c_void_p]),
    "sisic_unet_train_begin": (C.c_int, [C.c_void_p]),
    "sisic_unet_train_end": (C.c_int, [C.c_void_p]),
    "sisic_add_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int64, C.c_void_p]),
    "sisic_unet_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, c_int64_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p]),
    "sisic_mse_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "sisic_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sisic_unet_zero_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sisic_unet_optimizer_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float,
                                            C.POINTER(C.c_int), C.c_void_p]),
    "sisic_unet_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int64_p, c_float_p, c_float_p, C.c_int,
                                        C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float,
                                        C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]),
    "sisic_unet_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, C.c_int64]),
    "sisic_unet_train_steps": (C.c_int64, [C.c_void_p]),
    "sisic_conv2d_wgrad": (C.c_int, [C.c_void_p, C.POINTER(ConvArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sisic_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p]),
    "sisic_groupnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sisic_resnet_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "sisic_resnet_destroy": (C.c_int, [C.c_void_p]),
    "sisic_resnet_num_tensors": (C.c_int, [C.c_void_p]),
    "sisic_resnet_tensor_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "sisic_resnet_load": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), c_int64_p]),
    "sisic_resnet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "sisic_resnet_stem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sisic_resnet_input_gradient": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "sisic_resnet_gradcam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "sisic_resnet_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "sisic_unet_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "sisic_class_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "sisic_mask_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "sisic_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "sisic_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), c_int64_p, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sisic_profile_reset": (C.c_int, [C.c_void_p]),
}


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP library has not been built and synt_isic_amd has no CPU "
            "fallback.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C synt_isic_amd/csrc`).")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sisic_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsisic_hip.so ABI {lib.sisic_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


class SisicError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsisic_hip error {code}: {message}")
        self.code = code


def check(rc: int) -> None:
    if rc != SISIC_OK:
        msg = load().sisic_last_error()
        raise SisicError(rc, msg.decode("utf-8", "replace") if msg else "")
