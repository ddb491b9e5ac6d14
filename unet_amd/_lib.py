"""ctypes binding of ``libunet_mi355.so`` (the C ABI declared in ``include/unet_mi355.h``).

The library is loaded from ``unet_amd/lib``; it is never built at import time (run
``python -m unet_amd.build``).  There is no fallback: if the library is missing, every compute
entry point of the package raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libunet_mi355.so"
_lib = None


class MiUnetCfg(C.Structure):
    _fields_ = [
        ("spatial_dims", C.c_int32), ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("n_levels", C.c_int32), ("channels", C.c_int32 * 8), ("strides", C.c_int32 * 8),
        ("num_res_units", C.c_int32), ("dropout", C.c_float), ("dtype", C.c_int32), ("batch", C.c_int32),
        ("dims", C.c_int32 * 3), ("with_backward", C.c_int32),
    ]


class MiParamInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 160), ("ndim", C.c_int32), ("shape", C.c_int32 * 5), ("numel", C.c_int64),
        ("offset", C.c_int64), ("bucket", C.c_int32),
    ]


class MiStepInfo(C.Structure):
    _fields_ = [
        ("layer", C.c_char * 96), ("kernel", C.c_char * 96), ("work", C.c_double), ("work_is_flops", C.c_int32),
        ("bytes", C.c_double), ("total_ms", C.c_double), ("launches", C.c_int64),
    ]


class MiTrainStepArgs(C.Structure):
    _fields_ = [
        ("params", C.c_void_p), ("x", C.c_void_p), ("target", C.c_void_p), ("logits", C.c_void_p), ("dlogits", C.c_void_p),
        ("loss_sums", C.c_void_p), ("loss", C.c_void_p), ("loss_accum", C.c_void_p), ("grads", C.c_void_p),
        ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("workspace", C.c_void_p),
        ("target_dtype", C.c_int32), ("fold2", C.c_int32), ("smooth_nr", C.c_float), ("smooth_dr", C.c_float),
        ("lambda_dice", C.c_float), ("lambda_ce", C.c_float), ("do_adamw", C.c_int32), ("lr", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("grad_scale", C.c_float), ("step", C.c_int64), ("seed", C.c_uint64),
    ]


class MiNormInfo(C.Structure):
    _fields_ = [
        ("raw_offset", C.c_int64), ("pitch", C.c_int32), ("coff", C.c_int32), ("c", C.c_int32), ("dtype", C.c_int32),
        ("npix", C.c_int64), ("stats_offset", C.c_int64), ("layer", C.c_uint32),
    ]


class MiConvDesc(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("transposed", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32), ("nb", C.c_int32), ("in_dims", C.c_int32 * 3), ("dtype", C.c_int32),
    ]


class MiTensor(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p), ("pitch", C.c_int32), ("coff", C.c_int32),
        ("ptr2", C.c_void_p), ("pitch2", C.c_int32), ("coff2", C.c_int32), ("c1", C.c_int32),
    ]


_P = C.c_void_p
_I3 = C.c_int32 * 3

# name -> (restype, argtypes); every symbol declared in include/unet_mi355.h
PROTOTYPES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_int, []),
    "mi_unet_create": (C.c_int, [C.POINTER(MiUnetCfg), C.POINTER(_P)]),
    "mi_unet_destroy": (None, [_P]),
    "mi_unet_num_params": (C.c_int, [_P]),
    "mi_unet_param_info": (C.c_int, [_P, C.c_int, C.POINTER(MiParamInfo)]),
    "mi_unet_param_floats": (C.c_int64, [_P]),
    "mi_unet_workspace_bytes": (C.c_size_t, [_P]),
    "mi_unet_num_buckets": (C.c_int, [_P]),
    "mi_unet_bucket_range": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi_comm_unique_id": (C.c_int, [_P]),
    "mi_comm_init": (C.c_int, [C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "mi_comm_allreduce_bucket": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mi_comm_destroy": (C.c_int, [_P]),
    "mi_unet_flops": (C.c_double, [_P, C.c_int]),
    "mi_unet_num_steps": (C.c_int, [_P, C.c_int]),
    "mi_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "mi_unet_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_uint64, C.c_int, _P]),
    "mi_unet_backward_bucket": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_uint64, C.c_int, _P, _P]),
    "mi_unet_train_step": (C.c_int, [_P, C.POINTER(MiTrainStepArgs), _P]),
    "mi_unet_prof_enable": (C.c_int, [_P, C.c_int]),
    "mi_unet_prof_collect": (C.c_int, [_P]),
    "mi_unet_prof_reset": (C.c_int, [_P]),
    "mi_unet_step_info": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(MiStepInfo)]),
    "mi_unet_step_output_offset": (C.c_int64, [_P, C.c_int, C.c_int]),
    "mi_unet_step_aux_offset": (C.c_int64, [_P, C.c_int, C.c_int]),
    "mi_unet_num_norms": (C.c_int, [_P]),
    "mi_unet_norm_info": (C.c_int, [_P, C.c_int, C.POINTER(MiNormInfo)]),
    "mi_dice_ce_loss_fwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float,
                                      C.c_float, _P, _P, _P, _P]),
    "mi_dice_ce_loss_bwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float,
                                      C.c_float, _P, _P, C.c_float, _P, _P]),
    "mi_dice_ce_loss_fwd_noalloc": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float,
                                              C.c_float, _P, _P, _P]),
    "mi_dice_counts": (C.c_int, [_P, C.c_int, C.c_float, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "mi_fold_labels": (C.c_int, [_P, C.c_int, C.c_int64, _P]),
    "mi_adamw_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int64, C.c_float, _P]),
    "mi_sw_gather": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _I3, _I3, _I3, C.c_float, _P]),
    "mi_sw_blend": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _I3, _I3, _I3, _P]),
    "mi_fg_bbox": (C.c_int, [_P, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P]),
    "mi_pick_centers": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, _P, C.c_float, C.c_int, _P, _P]),
    "mi_crop_scale_rot": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "mi_conv_packed_bytes": (C.c_size_t, [C.POINTER(MiConvDesc)]),
    "mi_conv_pack": (C.c_int, [C.POINTER(MiConvDesc), C.c_int, _P, _P, _P]),
    "mi_conv_forward": (C.c_int, [C.POINTER(MiConvDesc), C.POINTER(MiTensor), _P, _P, C.POINTER(MiTensor),
                                  C.POINTER(MiTensor), _P, _P]),
    "mi_conv_dgrad": (C.c_int, [C.POINTER(MiConvDesc), C.POINTER(MiTensor), _P, C.POINTER(MiTensor),
                                C.POINTER(MiTensor), _P]),
    "mi_conv_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(MiConvDesc)]),
    "mi_conv_wgrad": (C.c_int, [C.POINTER(MiConvDesc), C.POINTER(MiTensor), C.POINTER(MiTensor), _P, _P, _P, _P]),
    "mi_conv1_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.POINTER(MiTensor), _P, _P]),
    "mi_conv1_transposed": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, C.POINTER(MiTensor), _P, _P, _P, _P, _P]),
    "mi_conv1_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, _P]),
    "mi_conv1_wgrad": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.POINTER(MiTensor), _P, _P, _P, _P]),
    "mi_instnorm_prelu_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(MiTensor), _P, _P,
                                        C.POINTER(MiTensor), C.POINTER(MiTensor), C.c_float, C.c_uint64, C.c_uint32, _P]),
    "mi_instnorm_prelu_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(MiTensor), C.POINTER(MiTensor),
                                        _P, _P, _P, _P, C.POINTER(MiTensor), C.c_float, C.c_uint64, C.c_uint32, _P]),
}


class LibraryMissing(RuntimeError):
    pass


def lib_path() -> Path:
    return Path(os.environ.get("UNET_AMD_LIB", str(_LIB_PATH)))


def load():
    """Loads the shared library once and attaches the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise LibraryMissing(
            f"{path} not found: build the HIP kernels first (python -m unet_amd.build). "
            "unet_amd has no CPU or eager fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mi_last_error()
        raise RuntimeError(f"libunet_mi355 {what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor, 0 for None."""
    return 0 if t is None else t.data_ptr()


def current_stream(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


LABEL_DTYPES = None


def label_dtype_code(t) -> int:
    import torch

    global LABEL_DTYPES
    if LABEL_DTYPES is None:
        LABEL_DTYPES = {torch.float32: 0, torch.uint8: 1, torch.int64: 2, torch.int32: 3}
    return LABEL_DTYPES.get(t.dtype, -1)
