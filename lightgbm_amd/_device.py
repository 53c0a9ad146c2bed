"""Device-side glue of the HIP core.

* Memory: every device buffer of the native core is a ``torch.uint8`` tensor
  handed out through an allocation hook, so PyTorch's caching allocator owns
  (and accounts for) all HBM the learners use.
* Stream: kernels and RCCL collectives are enqueued on PyTorch's current HIP
  stream, which orders them with any torch op that produced the inputs.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Dict

import numpy as np

from . import _lib

_live: Dict[int, "object"] = {}
_lock = threading.Lock()
_hooks = []   # keep the CFUNCTYPE objects alive


def _alloc(nbytes: int) -> int:
    import torch
    try:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
    except Exception:
        return 0
    p = t.data_ptr()
    with _lock:
        _live[p] = t
    return p


def _free(ptr) -> None:
    with _lock:
        _live.pop(int(ptr) if ptr else 0, None)


def install(lib) -> None:
    import torch
    torch.cuda.init()
    a = _lib.ALLOC_FN(_alloc)
    f = _lib.FREE_FN(_free)
    _hooks.extend([a, f])
    _lib.check(lib.lga_hip_set_allocator(a, f))
    sync_stream(lib)


def sync_stream(lib) -> None:
    """Points the native core at torch's current stream."""
    import torch
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.lga_hip_set_stream(ctypes.c_void_p(s)))


def device_bytes_in_use() -> int:
    with _lock:
        return sum(t.numel() for t in _live.values())


def predict(lib, booster_handle, mat, start_iteration: int, num_iteration: int, ptype: int, num_class: int = 1) -> np.ndarray:
    """Runs the LDS-tiled ensemble kernel; ``mat`` is a numpy array or a cuda tensor (row-major).
    Multiclass models return ``[rows, num_class]``."""
    import torch
    sync_stream(lib)
    if hasattr(mat, "data_ptr"):
        t = mat if mat.is_cuda else mat.cuda()
        if t.dtype not in (torch.float32, torch.float64):
            t = t.float()
        t = t.contiguous()
        out = torch.empty(t.shape[0] * num_class, dtype=torch.float64, device="cuda")
        _lib.check(lib.lga_hip_predict(booster_handle, ctypes.c_void_p(t.data_ptr()), 1,
                                       0 if t.dtype == torch.float32 else 1, t.shape[0], t.shape[1],
                                       start_iteration, num_iteration, ptype, ctypes.c_void_p(out.data_ptr()), 1))
        res = out.cpu().numpy()
        return res if num_class == 1 else res.reshape(t.shape[0], num_class)
    arr = np.ascontiguousarray(mat)
    out = np.empty(arr.shape[0] * num_class, dtype=np.float64)
    _lib.check(lib.lga_hip_predict(booster_handle, arr.ctypes.data_as(ctypes.c_void_p), 0,
                                   0 if arr.dtype == np.float32 else 1, arr.shape[0], arr.shape[1], start_iteration,
                                   num_iteration, ptype, out.ctypes.data_as(ctypes.c_void_p), 0))
    return out if num_class == 1 else out.reshape(arr.shape[0], num_class)
