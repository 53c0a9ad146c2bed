"""ctypes binding of the native core.

One of two in-tree libraries is loaded per process: ``_lga_hip.so`` when a GPU
is visible (it contains the host engine too; ``LGA_DEBUG_LIB=1`` picks its checked build
``_lga_hip_debug.so``), otherwise ``_lga_host.so``.
``LGA_FORCE_HOST=1`` forces the host library.  A missing library is built on
first use on CPU-only machines; on a GPU machine a missing HIP library is an
error (the kernels must be the code that runs, never a silent fallback).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p
from pathlib import Path

PKG = Path(__file__).resolve().parent

_lib = None
_is_hip = False


class LightGBMError(Exception):
    """Error raised by the native core."""


def _gpu_visible() -> bool:
    if os.environ.get("LGA_FORCE_HOST", "0") == "1":
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


ALLREDUCE_FN = ctypes.CFUNCTYPE(None, c_void_p, c_int64, c_int, c_int)
ALLGATHER_FN = ctypes.CFUNCTYPE(None, c_void_p, c_void_p, c_int64)
ALLOC_FN = ctypes.CFUNCTYPE(c_void_p, c_int64)
FREE_FN = ctypes.CFUNCTYPE(None, c_void_p)


def _declare(lib, hip: bool):
    P = POINTER
    sig = {
        "lga_last_error": (c_char_p, []),
        "lga_build_info": (c_char_p, []),
        "lga_alias_table_json": (c_char_p, []),
        "lga_resolve_alias": (c_int, [c_char_p, c_char_p, c_int]),
        "lga_dataset_from_mat": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_char_p, c_void_p, c_int, P(c_void_p)]),
        "lga_dataset_save_binary": (c_int, [c_void_p, c_char_p]),
        "lga_dataset_load_binary": (c_int, [c_char_p, P(c_void_p)]),
        "lga_dataset_from_sample": (c_int, [c_void_p, c_int64, c_int, c_int64, c_char_p, P(c_void_p)]),
        "lga_dataset_bin_matrix": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int]),
        "lga_dataset_from_reference": (c_int, [c_void_p, c_int64, P(c_void_p)]),
        "lga_dataset_subset": (c_int, [c_void_p, c_void_p, c_int64, P(c_void_p)]),
        "lga_dataset_set_field": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_int]),
        "lga_dataset_get_field": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, P(c_int64)]),
        "lga_dataset_num_data": (c_int, [c_void_p, P(c_int64)]),
        "lga_dataset_num_feature": (c_int, [c_void_p, P(c_int)]),
        "lga_dataset_num_used_feature": (c_int, [c_void_p, P(c_int)]),
        "lga_dataset_set_feature_names": (c_int, [c_void_p, c_char_p]),
        "lga_dataset_get_feature_names": (c_int, [c_void_p, c_void_p, c_int64, P(c_int64)]),
        "lga_dataset_bin_info": (c_int, [c_void_p, c_int, P(c_int), P(c_int), c_void_p, c_int, P(c_int)]),
        "lga_dataset_get_bins": (c_int, [c_void_p, c_void_p, c_void_p]),
        "lga_dataset_free": (c_int, [c_void_p]),
        "lga_comm_create_callback": (c_int, [c_int, c_int, ALLREDUCE_FN, ALLGATHER_FN, P(c_void_p)]),
        "lga_comm_free": (c_int, [c_void_p]),
        "lga_booster_create": (c_int, [c_void_p, c_char_p, c_void_p, P(c_void_p)]),
        "lga_booster_from_string": (c_int, [c_char_p, P(c_void_p)]),
        "lga_booster_free": (c_int, [c_void_p]),
        "lga_booster_add_valid": (c_int, [c_void_p, c_void_p, c_char_p]),
        "lga_booster_update": (c_int, [c_void_p, P(c_int)]),
        "lga_booster_update_custom": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, P(c_int)]),
        "lga_booster_rollback": (c_int, [c_void_p]),
        "lga_booster_set_leaf_value": (c_int, [c_void_p, c_int, c_int, c_double]),
        "lga_booster_shuffle": (c_int, [c_void_p, c_int, c_int, c_int64]),
        "lga_booster_refit": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_double, c_char_p]),
        "lga_booster_num_trees": (c_int, [c_void_p, P(c_int)]),
        "lga_booster_num_feature": (c_int, [c_void_p, P(c_int)]),
        "lga_booster_num_class": (c_int, [c_void_p, P(c_int)]),
        "lga_booster_eval_names": (c_int, [c_void_p, c_char_p, c_int]),
        "lga_booster_eval": (c_int, [c_void_p, c_int, c_void_p, c_int, P(c_int)]),
        "lga_booster_inner_nrows": (c_int, [c_void_p, c_int, P(c_int64)]),
        "lga_booster_inner_predict": (c_int, [c_void_p, c_int, c_int, c_void_p]),
        "lga_booster_get_gradients": (c_int, [c_void_p, c_void_p, c_void_p]),
        "lga_booster_predict": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
        "lga_booster_used_iterations": (c_int, [c_void_p, c_int, c_int, P(c_int)]),
        "lga_booster_save_string": (c_int, [c_void_p, c_int, c_int, c_int, P(c_char_p), P(c_int64)]),
        "lga_booster_dump_json": (c_int, [c_void_p, c_int, c_int, P(c_char_p), P(c_int64)]),
        "lga_booster_feature_importance": (c_int, [c_void_p, c_int, c_int, c_void_p]),
        "lga_booster_reset_parameter": (c_int, [c_void_p, c_char_p]),
        "lga_booster_prepend": (c_int, [c_void_p, c_void_p]),
        "lga_booster_stats": (c_int, [c_void_p, P(c_int), P(c_int), P(c_int)]),
        "lga_booster_profile_json": (c_int, [c_void_p, P(c_char_p), P(c_int64)]),
        "lga_booster_info": (c_int, [c_void_p, P(c_char_p), P(c_int), P(c_double)]),
    }
    if hip:
        sig.update({
            "lga_hip_set_allocator": (c_int, [ALLOC_FN, FREE_FN]),
            "lga_hip_set_stream": (c_int, [c_void_p]),
            "lga_hip_device_info": (c_int, [P(c_int), P(c_int64), P(c_int)]),
            "lga_hip_dataset_from_host_mat": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_char_p, P(c_void_p)]),
            "lga_hip_dataset_from_device_mat": (c_int, [c_void_p, c_int, c_int64, c_int, c_char_p, c_void_p, P(c_void_p)]),
            "lga_hip_dataset_upload_raw": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int]),
            "lga_hip_dataset_bin_device": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int]),
            "lga_hip_predict": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int]),
            "lga_hip_rccl_unique_id": (c_int, [c_void_p, c_int, P(c_int)]),
            "lga_hip_comm_create_rccl": (c_int, [c_int, c_int, c_void_p, c_int, P(c_void_p)]),
            "lga_hip_comm_create_loopback_group": (c_int, [c_int, P(c_void_p)]),
            "lga_hip_op": (c_int, [c_char_p, P(c_void_p), c_int, P(c_int64), c_int]),
        })
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def load():
    """Returns (ctypes library, is_hip)."""
    global _lib, _is_hip
    if _lib is not None:
        return _lib, _is_hip
    want_hip = _gpu_visible()
    # LGA_DEBUG_LIB=1: the -DLGA_DEBUG build of the HIP core, whose kernels check indices instead of faulting
    hip_name = "_lga_hip_debug.so" if os.environ.get("LGA_DEBUG_LIB", "0") == "1" else "_lga_hip.so"
    if os.environ.get("LGA_HIP_LIB"):      # development: another build of the HIP core next to this file (A/B timing)
        hip_name = os.environ["LGA_HIP_LIB"]
    path = PKG / (hip_name if want_hip else "_lga_host.so")
    if not path.exists():
        if want_hip:
            raise LightGBMError(
                f"{path} is missing on a GPU machine; build it first with `python -m lightgbm_amd.build`")
        from . import build
        build.build_host()
    mode = getattr(ctypes, "RTLD_GLOBAL", 0) if want_hip else 0
    lib = ctypes.CDLL(str(path), mode=mode)
    _declare(lib, want_hip)
    _lib, _is_hip = lib, want_hip
    if want_hip:
        from . import _device
        _device.install(lib)
    return _lib, _is_hip


def is_hip() -> bool:
    load()
    return _is_hip


def check(ret: int):
    if ret != 0:
        lib, _ = load()
        raise LightGBMError(lib.lga_last_error().decode("utf-8", "replace"))


def c_str(s: str) -> bytes:
    return s.encode("utf-8")
