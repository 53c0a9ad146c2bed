"""Single-kernel entry points (HIP build): run ONE training kernel on torch CUDA tensors.

The training loop never goes through these - it drives the kernels from C++ -
they exist so that every hot kernel can be checked against a plain NumPy /
PyTorch reference in isolation:

* :func:`hist_build`      - gradient histogram of a row list (LDS packed 64-bit atomics), feature-tile records
* :func:`lines_hist_build` - the same through the fused growth kernel's root pass over 128-byte row lines
* :func:`partition_rows`  - stable split of a row list by ``bin <= threshold``
* :func:`split_scan`      - best split of one leaf histogram (wave prefix scan + fp64 gain)
* :func:`pack_gradients`  - the (g, h) -> packed int64 quantisation used by the histograms

Without a GPU every function raises ``LightGBMError``.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Tuple

import numpy as np

from .. import _lib

TILE = 32
BINS = 256
QUANT_BITS = 16
SLOT_HEADER = 8


def _need_hip():
    lib, hip = _lib.load()
    if not hip:
        raise _lib.LightGBMError("lightgbm_amd.ops needs the HIP build and a visible GPU")
    from .. import _device
    _device.sync_stream(lib)
    return lib


def _op(lib, name: str, ptrs, ints):
    import torch
    P = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(p) for p in ptrs])
    I = (ctypes.c_int64 * len(ints))(*[int(i) for i in ints])
    _lib.check(lib.lga_hip_op(name.encode(), P, len(ptrs), I, len(ints)))
    torch.cuda.synchronize()


def _i32(values, device="cuda"):
    import torch
    return torch.tensor(np.asarray(values, dtype=np.int32), device=device)


def quant_exponent(vmax: float) -> int:
    """Largest k with vmax * 2^k < 2^16 (0 for vmax == 0)."""
    if not (vmax > 0):
        return 0
    m, e = np.frexp(np.float32(vmax))
    return QUANT_BITS - int(e)


def pack_gradients(g: np.ndarray, h: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """NumPy reference of the device quantisation: returns (packed int64, kg, kh)."""
    g = np.asarray(g, dtype=np.float32)
    h = np.asarray(h, dtype=np.float32)
    kg, kh = quant_exponent(float(np.abs(g).max(initial=0))), quant_exponent(float(h.max(initial=0)))
    lim = float(1 << QUANT_BITS)
    gq = np.rint(np.clip(np.ldexp(g, kg), -lim, lim)).astype(np.int64)
    hq = np.rint(np.clip(np.ldexp(h, kh), -lim, lim)).astype(np.int64)
    packed = (gq << 32) + hq
    return packed, kg, kh


def _slot_views(pool, fpad: int):
    """Views of one histogram slot: feature blocks [g 256 | h 256], then the header (rows, sum g, sum h)."""
    blocks = pool[:fpad * 2 * BINS].view(fpad, 2, BINS)
    return blocks[:, 0, :], blocks[:, 1, :], pool[fpad * 2 * BINS:]


def hist_build(bins_rt, rows, packed, num_features: int, rows_per_chunk: int = 4096):
    """bins_rt uint8 [tiles, N, 32], rows int32 [n], packed int64 [N] -> (g int64 [tiles*32, 256], h, header[8]).
    Runs the feature-tile histogram kernel (data with several tiles, feature-parallel)."""
    import torch
    lib = _need_hip()
    T, N = int(bins_rt.shape[0]), int(bins_rt.shape[1])
    n = int(rows.numel())
    fpad = T * TILE
    slot_words = fpad * 2 * BINS + SLOT_HEADER
    nch = max(1, -(-n // rows_per_chunk))
    task = np.zeros(17, dtype=np.int32)
    task[13] = nch   # num_hchunk; left_node = 0, hist_child = 0, new_slot = 0, first_hchunk = 0
    tasks = _i32(task)
    chunk_task = torch.zeros(nch, dtype=torch.int32, device="cuda")
    nodes = _i32([0, n, 0, 0])
    pool = torch.zeros(slot_words, dtype=torch.int64, device="cuda")
    table = torch.zeros(2, dtype=torch.int64, device="cuda")
    rows = rows.contiguous()
    _op(lib, "histogram",
        [tasks.data_ptr(), chunk_task.data_ptr(), nodes.data_ptr(), rows.data_ptr(), rows.data_ptr(),
         bins_rt.data_ptr(), packed.data_ptr(), pool.data_ptr(), table.data_ptr()],
        [nch, rows_per_chunk, N, num_features, 0, 1, T, T])
    return _slot_views(pool, fpad)


def lines_hist_build(bins_rt, rows, num_features: int, packed=None, g=None, h=None, unit_rows: int = 4096):
    """Root pass of the fused growth kernel over row lines (one feature tile), histogram only.

    bins_rt uint8 [1, N, 32], rows int32 [n].  Either ``packed`` int64 [N] (quantised gradients) or float32
    ``g`` / ``h`` [N], which the pass quantises itself (scales from their maxima over ``rows``).
    Returns (g int64 [32, 256], h, header[8], (kg, kh) or None)."""
    import torch
    lib = _need_hip()
    assert int(bins_rt.shape[0]) == 1 and num_features <= TILE
    N = int(bins_rt.shape[1])
    n = int(rows.numel())
    slot_words = TILE * 2 * BINS + SLOT_HEADER
    nodes = _i32([0, n, 0, 0])
    pool = torch.zeros(slot_words, dtype=torch.int64, device="cuda")
    table = torch.zeros(2, dtype=torch.int64, device="cuda")
    tickets = torch.zeros(4, dtype=torch.int32, device="cuda")
    lines = torch.zeros(((N + 2) // 3) * 128, dtype=torch.uint8, device="cuda")
    rows = rows.contiguous()
    qs = torch.zeros(2, dtype=torch.int32, device="cuda")
    if packed is not None:
        ptrs = [nodes.data_ptr(), rows.data_ptr(), bins_rt.data_ptr(), packed.data_ptr(), lines.data_ptr(),
                pool.data_ptr(), table.data_ptr(), tickets.data_ptr(), 0, 0, 0, 0]
    else:
        gm = float(g[rows.long()].abs().max()) if n else 0.0
        hm = float(h[rows.long()].max()) if n else 0.0
        maxbits = torch.tensor(np.array([gm, hm], dtype=np.float32).view(np.int32), device="cuda")
        ptrs = [nodes.data_ptr(), rows.data_ptr(), bins_rt.data_ptr(), 0, lines.data_ptr(), pool.data_ptr(),
                table.data_ptr(), tickets.data_ptr(), g.data_ptr(), h.data_ptr(), maxbits.data_ptr(), qs.data_ptr()]
    _op(lib, "lines_histogram", ptrs, [n, N, num_features, unit_rows])
    G, H, hd = _slot_views(pool, TILE)
    scales = None if packed is not None else tuple(int(v) for v in qs.cpu().numpy())
    return G, H, hd, scales


def partition_rows_single_pass(rows, bins_fm, feature: int, threshold: int, missing_bin: int = -1, default_left: bool = False,
                               block_rows: int = 2048):
    """The growth rounds' single-pass partition of ``rows`` (int32 [n]) by bins_fm[feature, row] <= threshold: lefts at
    the front, rights behind them, in no particular order inside either.  Returns (new rows, n_left)."""
    import torch
    lib = _need_hip()
    n = int(rows.numel())
    N = int(bins_fm.shape[1])
    nch = max(1, -(-n // block_rows))
    task = np.zeros(17, dtype=np.int32)
    task[0], task[1], task[2] = 0, 1, 2
    task[3], task[4], task[5], task[6] = feature, threshold, int(default_left), missing_bin
    task[10], task[11] = 0, nch
    tasks = _i32(task)
    nodes = _i32([0, n, 0, 0] + [0] * 8)
    src = rows.contiguous().clone()
    dst = torch.full_like(src, -1)
    cursor = torch.zeros(32, dtype=torch.int64, device="cuda")
    task_left = torch.zeros(1, dtype=torch.int32, device="cuda")
    _op(lib, "grow_partition",
        [tasks.data_ptr(), nodes.data_ptr(), src.data_ptr(), dst.data_ptr(), bins_fm.data_ptr(), cursor.data_ptr(), task_left.data_ptr()],
        [nch, N, block_rows])
    torch.cuda.synchronize()
    nd = nodes.cpu().numpy().reshape(3, 4)
    nl = int(task_left.item())
    assert nd[1, 1] == nl and nd[2, 0] == nl and nd[2, 1] == n - nl
    assert int(cursor[0].item()) == 0          # the last block puts the cursor back for the next round
    return dst, nl


def partition_rows(rows, bins_fm, feature: int, threshold: int, missing_bin: int = -1, default_left: bool = False):
    """Stable partition of ``rows`` (int32 [n]) by bins_fm[feature, row] <= threshold.  Returns (new rows, n_left)."""
    import torch
    lib = _need_hip()
    n = int(rows.numel())
    N = int(bins_fm.shape[1])
    nch = max(1, -(-n // 2048))
    task = np.zeros(17, dtype=np.int32)
    task[0], task[1], task[2] = 0, 1, 2
    task[3], task[4], task[5], task[6] = feature, threshold, int(default_left), missing_bin
    task[10], task[11] = 0, nch
    tasks = _i32(task)
    chunk_task = torch.zeros(nch, dtype=torch.int32, device="cuda")
    nodes = _i32([0, n, 0, 0] + [0] * 8)
    src = rows.contiguous().clone()
    dst = torch.full_like(src, -1)
    mask = torch.zeros(nch * 32, dtype=torch.int64, device="cuda")
    chunk_left = torch.zeros(nch, dtype=torch.int32, device="cuda")
    task_left = torch.zeros(1, dtype=torch.int32, device="cuda")
    _op(lib, "partition",
        [tasks.data_ptr(), chunk_task.data_ptr(), nodes.data_ptr(), src.data_ptr(), dst.data_ptr(), bins_fm.data_ptr(),
         mask.data_ptr(), chunk_left.data_ptr(), task_left.data_ptr()], [nch, N])
    nd = nodes.cpu().numpy().reshape(3, 4)
    nl = int(task_left.item())
    assert nd[1, 1] == nl and nd[2, 0] == nl and nd[2, 1] == n - nl
    return dst, nl


class _SplitParams(ctypes.Structure):
    _fields_ = [("lambda_l1", ctypes.c_double), ("lambda_l2", ctypes.c_double), ("min_gain_to_split", ctypes.c_double),
                ("min_sum_hessian_in_leaf", ctypes.c_double), ("max_delta_step", ctypes.c_double),
                ("min_data_in_leaf", ctypes.c_int32), ("max_depth", ctypes.c_int32),
                ("bynode_fraction", ctypes.c_double), ("bynode_seed", ctypes.c_uint64),
                ("extra_trees", ctypes.c_int32), ("mask_words", ctypes.c_int32), ("extra_seed", ctypes.c_uint64),
                ("node_bounds", ctypes.c_void_p), ("node_masks", ctypes.c_void_p), ("path_smooth", ctypes.c_double),
                ("feature_scale", ctypes.c_void_p), ("feature_offset", ctypes.c_void_p), ("split_penalty", ctypes.c_double)]


def split_scan(g, h, count: int, total_g: int, total_h: int, num_bins, missing_bins, kg: int, kh: int,
               lambda_l1: float = 0.0, lambda_l2: float = 0.0, min_gain_to_split: float = 0.0,
               min_sum_hessian_in_leaf: float = 1e-3, min_data_in_leaf: int = 20) -> Dict[str, Any]:
    """g, h int64 [F_pad, 256] (F_pad a multiple of 32); returns the winning split as a dict."""
    import torch
    lib = _need_hip()
    F = len(num_bins)
    fpad = int(g.shape[0])
    chunk_words = fpad * 2 * BINS + SLOT_HEADER
    pool = torch.zeros(chunk_words, dtype=torch.int64, device="cuda")
    blocks = pool[:fpad * 2 * BINS].view(fpad, 2, BINS)
    blocks[:, 0, :] = g
    blocks[:, 1, :] = h
    hd = fpad * 2 * BINS
    pool[hd + 0], pool[hd + 1], pool[hd + 2] = int(count), int(total_g), int(total_h)
    table = torch.tensor([0, chunk_words], dtype=torch.int64, device="cuda")
    meta = np.zeros((F, 4), dtype=np.int32)
    meta[:, 0], meta[:, 1], meta[:, 2] = num_bins, missing_bins, 1
    meta_t = _i32(meta.reshape(-1))
    totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    scan_totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    qs = _i32([kg, kh])
    sp = _SplitParams(lambda_l1, lambda_l2, min_gain_to_split, min_sum_hessian_in_leaf, 0.0, min_data_in_leaf, -1, 1.0, 0, 0, 0, 0, None, None, 0.0, None, None, 0.0)
    n_blocks = max(1, (F + 63) // 64)
    block_best = torch.zeros(n_blocks * 4, dtype=torch.int64, device="cuda")
    counters = torch.zeros(16, dtype=torch.int32, device="cuda")
    out = torch.zeros(8 + 9 + 2, dtype=torch.int64, device="cuda")    # flag block (64 B) | SplitInfo (72 B) | QuantScale
    results = torch.zeros(9, dtype=torch.int64, device="cuda")
    _op(lib, "scan", [meta_t.data_ptr(), pool.data_ptr(), table.data_ptr(), totals.data_ptr(), qs.data_ptr(),
                      ctypes.addressof(sp), block_best.data_ptr(), counters.data_ptr(), scan_totals.data_ptr(),
                      out.data_ptr(), results.data_ptr()], [F, fpad])
    raw = out.cpu().numpy()
    assert int(raw[:1].view(np.uint32)[0]) == 1, "the split search did not raise its completion flag"
    raw = raw[8:17]
    ints = raw[1:3].view(np.int32)
    return {"gain": float(raw[:1].view(np.float64)[0]), "feature": int(ints[0]), "threshold": int(ints[1]),
            "default_left": int(ints[2]), "left_g": int(raw[3]), "left_h": int(raw[4]), "total_g": int(raw[5]),
            "total_h": int(raw[6]), "left_cnt": int(raw[7]), "total_cnt": int(raw[8])}
