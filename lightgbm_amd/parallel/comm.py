"""Communicators for the distributed tree learners.

* :class:`RcclComm` - the production path: one process per GPU, RCCL over xGMI.
  The native core binds librccl itself and enqueues every collective on the
  learner's HIP stream; ``torch.distributed`` is only used once, to hand
  rank 0's RCCL unique id to the other ranks.
* :class:`TorchComm` - collectives through ``torch.distributed`` on host buffers
  (``gloo``): used by the CPU engine and by the multi-process CPU tests.
* :class:`ThreadComm` - an in-process loop-back communicator: P learners run in
  P threads of one process and reduce through shared host buffers.  It tests
  the sharding logic without any process group.

Payloads per growth round (data-parallel, device engine): ONE in-place integer
reduce-scatter over the round's new histogram slots - the pool lays a round's
slots out rank-chunk-major, every chunk ending in a copy of the slot headers
(row counts, gradient totals) - after which each rank searches only the
features it owns, and ONE all-gather of the ranks' winners (32 bytes per
child).  That is two collectives per round, i.e. about 2 x tree depth per tree
instead of two per leaf; a rank receives 1/P of the histogram bytes.  Integer
sums make the result independent of rank order and bit-identical to serial
training.  Feature-parallel exchanges only the per-leaf winners.  Host-only
communicators (the two below) serve reduce-scatter as an all-reduce.
"""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional

import numpy as np

from .. import _lib

_DT = {0: np.int64, 1: np.float64, 2: np.float32}


class _CallbackComm:
    """Base: wraps Python collectives into the native callback communicator."""

    def __init__(self, rank: int, size: int):
        lib, _ = _lib.load()
        self.rank, self.size = rank, size
        self._ar = _lib.ALLREDUCE_FN(self._allreduce)
        self._ag = _lib.ALLGATHER_FN(self._allgather)
        h = ctypes.c_void_p()
        _lib.check(lib.lga_comm_create_callback(rank, size, self._ar, self._ag, ctypes.byref(h)))
        self.handle = h

    # buffers arrive as raw pointers
    def _allreduce(self, buf, n, dtype, op):
        arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_byte)), shape=(n * np.dtype(_DT[dtype]).itemsize,))
        view = arr.view(_DT[dtype])
        self.allreduce(view, "max" if op == 1 else "sum")

    def _allgather(self, inp, out, nbytes):
        a = np.ctypeslib.as_array(ctypes.cast(inp, ctypes.POINTER(ctypes.c_ubyte)), shape=(nbytes,))
        o = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_ubyte)), shape=(nbytes * self.size,))
        self.allgather(a, o)

    def allreduce(self, view: np.ndarray, op: str):
        raise NotImplementedError

    def allgather(self, a: np.ndarray, out: np.ndarray):
        raise NotImplementedError

    def close(self):
        if getattr(self, "handle", None):
            lib, _ = _lib.load()
            lib.lga_comm_free(self.handle)
            self.handle = None


class TorchComm(_CallbackComm):
    """Host-buffer collectives through a ``torch.distributed`` process group (gloo; under an ``nccl`` group
    the buffers take a detour through device tensors)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self._group = group
        self._on_device = dist.get_backend(group) == "nccl"
        super().__init__(dist.get_rank(group), dist.get_world_size(group))

    def allreduce(self, view, op):
        import torch
        t = torch.from_numpy(view)
        red = self._dist.ReduceOp.MAX if op == "max" else self._dist.ReduceOp.SUM
        if self._on_device:
            d = t.cuda()
            self._dist.all_reduce(d, op=red, group=self._group)
            t.copy_(d)
        else:
            self._dist.all_reduce(t, op=red, group=self._group)

    def allgather(self, a, out):
        import torch
        dev = "cuda" if self._on_device else "cpu"
        outs = [torch.empty(len(a), dtype=torch.uint8, device=dev) for _ in range(self.size)]
        self._dist.all_gather(outs, torch.from_numpy(a.copy()).to(dev), group=self._group)
        out[:] = torch.cat(outs).cpu().numpy()


class _ThreadHub:
    def __init__(self, size: int):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots: List[Optional[np.ndarray]] = [None] * size


class ThreadComm(_CallbackComm):
    """Loop-back communicator: create one hub, then one ThreadComm per rank (each used from its own thread)."""

    @staticmethod
    def make_group(size: int) -> List["ThreadComm"]:
        hub = _ThreadHub(size)
        return [ThreadComm(hub, r) for r in range(size)]

    def __init__(self, hub: _ThreadHub, rank: int):
        self._hub = hub
        super().__init__(rank, hub.size)

    def allreduce(self, view, op):
        hub = self._hub
        hub.slots[self.rank] = view.copy()
        hub.barrier.wait()
        acc = hub.slots[0].copy()
        for r in range(1, hub.size):       # fixed rank order
            acc = np.maximum(acc, hub.slots[r]) if op == "max" else acc + hub.slots[r]
        hub.barrier.wait()
        view[:] = acc

    def allgather(self, a, out):
        hub = self._hub
        hub.slots[self.rank] = a.copy()
        hub.barrier.wait()
        out[:] = np.concatenate([hub.slots[r] for r in range(hub.size)])
        hub.barrier.wait()


def _share_unique_id(group, make_id):
    """Rank 0 creates a communicator id, every rank receives it.  All ranks run the SAME collective (one
    broadcast of id + status byte) whether or not rank 0 succeeded, and all ranks learn about a failure: returns
    (id bytes, None) everywhere, or (None, error text) everywhere."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    buf = np.zeros(129, dtype=np.uint8)
    err = None
    if rank == 0:
        try:
            buf[:128] = make_id()
            buf[128] = 1
        except Exception as e:      # noqa: BLE001 - reported to every rank below
            err = str(e)
    t = torch.from_numpy(buf)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    buf = t.cpu().numpy()
    if buf[128] != 1:
        return None, err or "rank 0 could not create a communicator id"
    return buf[:128].copy(), None


class RcclComm:
    """RCCL communicator owned by the native core (HIP build only).

    The host driver of some machines shares device memory between processes through dmabuf only; RCCL then
    needs ``HSA_ENABLE_IPC_MODE_LEGACY=0`` in the environment BEFORE the HIP runtime starts.  Importing
    ``lightgbm_amd`` sets it (unless the variable is already set), so import the package before the first GPU call."""

    def __init__(self, group=None):
        import torch.distributed as dist
        lib, hip = _lib.load()
        if not hip:
            raise _lib.LightGBMError("RcclComm needs the HIP build and a visible GPU")
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)

        def make_id():
            uid = np.zeros(128, dtype=np.uint8)
            n = ctypes.c_int()
            _lib.check(lib.lga_hip_rccl_unique_id(uid.ctypes.data_as(ctypes.c_void_p), 128, ctypes.byref(n)))
            return uid
        uid, err = _share_unique_id(group, make_id)
        if uid is None:
            raise _lib.LightGBMError(f"no RCCL unique id: {err}")
        from .. import _device
        _device.sync_stream(lib)
        h = ctypes.c_void_p()
        _lib.check(lib.lga_hip_comm_create_rccl(self.rank, self.size, uid.ctypes.data_as(ctypes.c_void_p), 128,
                                                ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib, _ = _lib.load()
            lib.lga_comm_free(self.handle)
            self.handle = None


class DeviceLoopbackComm:
    """In-process loop-back group for the HIP engine: P learners in P threads on ONE GPU (tests only)."""

    def __init__(self, handle, rank, size):
        self.handle, self.rank, self.size = handle, rank, size

    @staticmethod
    def make_group(size: int) -> List["DeviceLoopbackComm"]:
        lib, hip = _lib.load()
        if not hip:
            raise _lib.LightGBMError("DeviceLoopbackComm needs the HIP build and a visible GPU")
        arr = (ctypes.c_void_p * size)()
        _lib.check(lib.lga_hip_comm_create_loopback_group(size, arr))
        return [DeviceLoopbackComm(ctypes.c_void_p(arr[r]), r, size) for r in range(size)]

    def close(self):
        if getattr(self, "handle", None):
            lib, _ = _lib.load()
            lib.lga_comm_free(self.handle)
            self.handle = None


def _native_or_fallback(group, make_native):
    """Creates the native communicator on every rank; should that fail on ANY rank, every rank falls back to the
    process group's own collectives (slower: buffers are staged through host memory) instead of leaving the job
    without a communicator.  Every rank runs the same collectives on every path: those inside ``make_native``
    (see :func:`_share_unique_id`) and one MIN all-reduce of the outcome."""
    import torch
    import torch.distributed as dist
    comm, err = None, None
    try:
        comm = make_native(group)
    except Exception as e:      # noqa: BLE001 - any failure takes the fallback
        err = e
    ok = torch.tensor([1 if comm is not None else 0], device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 1:
        return comm
    if comm is not None:
        comm.close()
    import warnings
    warnings.warn(f"native RCCL communicator unavailable ({err}); using torch.distributed collectives")
    return TorchComm(group)


def default_comm(group=None):
    """RCCL when training on GPUs under an ``nccl`` process group, torch.distributed host collectives otherwise
    (CPU engine, or a ``gloo`` group used to rehearse a multi-rank run on fewer GPUs than ranks: the device
    engine then stages its buffers through host memory); None when not distributed."""
    import os
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    if dist.get_world_size(group) <= 1:
        # LGA_DP_KEEP_SINGLE_RANK=1: a communicator of one rank, which the device engine then really trains
        # through (a way to time the distributed code path on a single GPU)
        if os.environ.get("LGA_DP_KEEP_SINGLE_RANK") == "1" and _lib.is_hip() and dist.get_backend(group) == "nccl":
            return RcclComm(group)
        return None
    if _lib.is_hip() and dist.get_backend(group) == "nccl":
        return _native_or_fallback(group, RcclComm)
    return TorchComm(group)
