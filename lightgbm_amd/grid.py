"""Parameter-grid cross-validation search: the headline workload of the reference.

Mirrors r/gridsearchCV.R:92-144:

* :func:`expand_grid`       <- ``expand.grid(...)`` (R:92-100): Cartesian product, the FIRST
  factor varies fastest, result columns ``iteration`` / ``score`` pre-allocated as NaN.
* :func:`grid_search_cv`    <- the ``for`` loop over grid rows running ``lgb.cv`` (R:104-119),
  re-organised as a work queue: rows are ordered most-expensive-first and pulled by one worker
  process per GPU (or CPU worker); every finished row is appended to a durable JSON-lines
  checkpoint (fsync per row), the analogue of ``save(paramGrid, ...)`` after every row (R:118);
  a restarted search skips finished rows (R:121 ``load("paramGrid.RData")``); a worker that dies
  hands its in-flight rows back to the queue.  On the HIP build a worker keeps SEVERAL rows in flight
  (``rows_in_flight``): one 5-fold CV of a 37 k-row table is a chain of tiny latency-bound kernels, so
  rows x folds host threads, each on its own HIP stream, are what fills the device.
* :func:`grid_search_cv_distributed` <- the same loop under ``torch.distributed``: every rank pulls the
  next row from a counter in the rendezvous store, writes its own checkpoint, and the tables are
  all-gathered; a restart merges every rank's checkpoint first.
* :func:`rank_grid`         <- ``paramGrid[order(-paramGrid$score), ]`` (R:122).
* :func:`refit_top` / :func:`average_predictions` <- refit the best m rows with their CV-chosen
  number of rounds and average the predictions (R:124-144).
"""
from __future__ import annotations

import hashlib
import itertools
import json
import os
import time
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["expand_grid", "grid_search_cv", "grid_search_cv_distributed", "rank_grid", "refit_top",
           "average_predictions", "load_checkpoint", "GridCheckpoint"]

RESULT_COLUMNS = ("iteration", "score")


def expand_grid(**factors):
    """Cartesian product as a ``pandas.DataFrame``; the first factor varies fastest (R's ``expand.grid``).

    Scalars are constants carried as columns (``bagging_freq = 4``, ``nthread = 4``); a factor given
    as ``None`` becomes a NaN result column (``iteration = NA_integer_``, ``score = NA_real_``).
    """
    import pandas as pd
    names = list(factors.keys())
    levels = []
    for n in names:
        v = factors[n]
        if v is None:
            levels.append([np.nan])
        elif isinstance(v, (str, bytes)) or not hasattr(v, "__iter__"):
            levels.append([v])
        else:
            levels.append(list(v))
    # itertools.product varies the LAST factor fastest -> reverse
    rows = [tuple(reversed(c)) for c in itertools.product(*reversed(levels))]
    return pd.DataFrame(rows, columns=names)


def _row_params(row: Dict[str, Any]) -> Dict[str, Any]:
    from .rstyle import _row_to_params
    return _row_to_params({k: v for k, v in row.items() if k not in RESULT_COLUMNS})


def _params_hash(params: Dict[str, Any], extra: Dict[str, Any]) -> str:
    blob = json.dumps({"p": {k: params[k] for k in sorted(params)}, "x": {k: extra[k] for k in sorted(extra)}},
                      sort_keys=True, default=str)
    return hashlib.sha1(blob.encode()).hexdigest()[:16]


class GridCheckpoint:
    """Append-only JSON-lines log of finished grid rows; every append is flushed and fsync'ed."""

    def __init__(self, path: Optional[str]):
        import threading
        self.path = path
        self.done: Dict[int, Dict[str, Any]] = {}
        self._lock = threading.Lock()        # rows finish on several threads
        if path and os.path.exists(path):
            with open(path, "r") as f:
                for line in f:
                    line = line.strip()
                    if not line:
                        continue
                    try:
                        rec = json.loads(line)
                    except json.JSONDecodeError:
                        continue    # a torn last line of a killed run
                    if rec.get("status") == "ok":
                        self.done[int(rec["row"])] = rec

    def has(self, row: int, phash: str) -> bool:
        return row in self.done and self.done[row].get("hash") == phash

    def append(self, rec: Dict[str, Any]):
        with self._lock:
            self.done[int(rec["row"])] = rec
            if not self.path:
                return
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")
                f.flush()
                os.fsync(f.fileno())


def load_checkpoint(path: str) -> Dict[int, Dict[str, Any]]:
    return GridCheckpoint(path).done


def _cost_estimate(p: Dict[str, Any]) -> float:
    """Relative cost of a grid row: small learning rates run the most rounds, big trees cost more per round."""
    lr = float(p.get("learning_rate", 0.1))
    leaves = float(p.get("num_leaves", 31))
    return leaves ** 0.5 / max(lr, 1e-6)


def _cv_one(params: Dict[str, Any], dtrain, cv_kwargs: Dict[str, Any]) -> Tuple[int, float]:
    from .rstyle import lgb_cv
    cvm = lgb_cv(params, dtrain, **cv_kwargs)
    return int(cvm.best_iter), float(cvm.best_score)


def _cv_one_steps(params: Dict[str, Any], dtrain, cv_kwargs: Dict[str, Any]):
    """:func:`_cv_one` as a generator of lockstep requests (rstyle.lgb_cv_steps)."""
    from .rstyle import lgb_cv_steps
    cvm = yield from lgb_cv_steps(params, dtrain, rollback_overrun=False, **cv_kwargs)   # the boosters are dropped
    return int(cvm.best_iter), float(cvm.best_score)


def _run_rows_batched(next_row: Callable[[], Optional[int]], start_row: Callable[[int], Any],
                      finish_row: Callable[[int, Any, float], None], width: int):
    """Continuous batching of grid rows on one device: up to ``width`` rows are alive at a time, each a generator
    that asks for its fold boosters to be advanced a few iterations (engine.cv_steps); the requests of all living
    rows are answered by ONE native call - on the HIP engine one kernel launch with one workgroup per booster - and a
    row that ends (early stop, rounds used up) hands its place to the next one.  A row that never asks (it ran in
    the plain per-iteration loop) simply finishes when started."""
    from .basic import Booster
    alive: Dict[int, List[Any]] = {}      # row -> [generator, pending request, start time]
    exhausted = False
    trace = os.environ.get("LGA_GRID_TRACE")
    t_start = t_native = t_rows = t_first = 0.0
    n_calls = n_booster_iters = 0
    while True:
        t_a = time.perf_counter()
        while not exhausted and len(alive) < width:
            i = next_row()
            if i is None:
                exhausted = True
                break
            t0 = time.time()
            gen = start_row(i)
            t_gen = time.perf_counter()
            try:
                alive[i] = [gen, next(gen), t0]
            except StopIteration as stop:
                finish_row(i, stop.value, time.time() - t0)
            t_first += time.perf_counter() - t_gen
        if not alive:
            if trace:
                print(f"[grid trace] {n_calls} native calls, {n_booster_iters} booster iterations: starting rows {t_start:.2f}s, "
                      f"(of which up to the rows' first request {t_first:.2f}s), native calls {t_native:.2f}s, per-row processing {t_rows:.2f}s", flush=True)
            return
        ids = list(alive)
        n_iters = min(alive[i][1][1] for i in ids)
        boosters = [b for i in ids for b in alive[i][1][0]]
        t_b = time.perf_counter()
        values, no_split, done = Booster.update_group(boosters, n_iters, True)
        t_c = time.perf_counter()
        t_start += t_b - t_a
        t_native += t_c - t_b
        n_calls += 1
        n_booster_iters += done * len(boosters)
        at = 0
        for i in ids:
            gen, (bs, _), t0 = alive[i]
            part = (values[:, at: at + len(bs)], no_split[:, at: at + len(bs)], done)
            at += len(bs)
            try:
                alive[i][1] = gen.send(part)
            except StopIteration as stop:
                del alive[i]
                finish_row(i, stop.value, time.time() - t0)
        t_rows += time.perf_counter() - t_c


def _auto_rows_in_flight(dtrain, rows: Sequence[Dict[str, Any]]) -> int:
    """Rows one process keeps in flight.  HIP engine: 48 for tables of at most 65536 rows - their folds run as
    one-workgroup boosters, 48 rows x 5 folds fill the 256 compute units - 4 up to 500 k rows (bigger folds fill
    the device by themselves); 1 on the CPU engine, whose histogram loops already use every core."""
    from . import _lib
    env = os.environ.get("LGA_GRID_ROWS_IN_FLIGHT")
    if env:
        return max(1, int(env))
    if not _lib.is_hip():
        return 1
    where = [getattr(dtrain, "params", None) or {}] + list(rows)
    if any(str(r.get("device_type", r.get("device", "gpu"))).lower() in ("cpu", "host") for r in where):
        return 1
    if dtrain.num_data() <= 65536:
        return 48
    return 4 if dtrain.num_data() <= 500_000 else 1


def _callers_device():
    """The GPU the calling thread works on (None: CPU build or no GPU).  A thread started later does not inherit it:
    the current device is a per-thread setting that starts at device 0."""
    from . import _lib
    if not _lib.is_hip():
        return None
    import torch
    return torch.cuda.current_device() if torch.cuda.is_available() else None


def _own_stream(device):
    """Context of one row thread: on the HIP build the thread's boosters live on a stream of their own - on
    ``device``, the GPU of the thread that started the search - so that the rows in flight overlap on the device
    instead of queueing on the default stream."""
    import contextlib
    if device is None:
        return contextlib.nullcontext()
    import torch
    return torch.cuda.stream(torch.cuda.Stream(device=device))   # entering it makes `device` this thread's device


def _run_rows(next_row: Callable[[], Optional[int]], run_one: Callable[[int], None], in_flight: int):
    """Drains a row source with ``in_flight`` host threads; the first exception stops the pulls and is re-raised
    after the rows already started have finished (their results are durable by then)."""
    if in_flight <= 1:
        while True:
            i = next_row()
            if i is None:
                return
            run_one(i)
    import threading
    from concurrent.futures import ThreadPoolExecutor
    failed: List[BaseException] = []
    gate = threading.Lock()

    device = _callers_device()

    def drain():
        with _own_stream(device):
            _drain()

    def _drain():
        while not failed:
            with gate:
                i = next_row()
            if i is None:
                return
            try:
                run_one(i)
            except BaseException as e:          # noqa: BLE001 - re-raised by the caller's thread
                failed.append(e)
                return

    with ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix="lga-grid") as pool:
        for f in [pool.submit(drain) for _ in range(in_flight)]:
            f.result()
    if failed:
        raise failed[0]


def _drain_rows(next_row: Callable[[], Optional[int]], rows, dtrain, cv_kwargs, on_start: Callable[[int], None],
                on_done: Callable[[int, int, float, float], None], in_flight: Optional[int], batch: Optional[bool]):
    """Evaluates grid rows from ``next_row`` until it returns None.  batch (None: HIP build and at most 65536 rows):
    one thread batches the living rows' boosters into single native calls (_run_rows_batched); otherwise
    ``in_flight`` host threads run one row each (_run_rows)."""
    from . import _lib
    in_flight = in_flight or _auto_rows_in_flight(dtrain, rows)
    if batch is None:
        batch = in_flight > 1 and _lib.is_hip() and dtrain.num_data() <= 65536 and _auto_rows_in_flight(dtrain, rows) > 1
    if batch:
        def start_row(i):
            on_start(i)
            return _cv_one_steps(rows[i], dtrain, cv_kwargs)

        _run_rows_batched(next_row, start_row, lambda i, res, secs: on_done(i, res[0], res[1], secs), in_flight)
    else:
        def run_one(i):
            on_start(i)
            t0 = time.time()
            it, sc = _cv_one(rows[i], dtrain, cv_kwargs)
            on_done(i, it, sc, time.time() - t0)

        _run_rows(next_row, run_one, in_flight)


def _worker_main(worker_id: int, device: Optional[str], X, y, weight, dataset_params, rows, cv_kwargs, conn,
                 fail_after: Optional[int], rows_in_flight: Optional[int] = None, batch_rows: Optional[bool] = None):
    """Worker process.  It talks to the scheduler over a pipe of its own - ("want",) is answered with a row id or
    None (stop), ("start", row), ("done", row, (iteration, score, seconds)), ("error", text), ("exit",) are
    reports - so a worker that dies, whenever and however, takes no lock with it that another process waits for
    (a queue shared by all workers would: its write lock lives in shared memory)."""
    if device is not None and device != "cpu":
        os.environ["HIP_VISIBLE_DEVICES"] = str(device)
        os.environ["CUDA_VISIBLE_DEVICES"] = str(device)
    elif device == "cpu":
        os.environ["LGA_FORCE_HOST"] = "1"
    import threading
    from .basic import Dataset
    send = threading.Lock()   # the row threads of this process share the pipe: one message at a time ...
    ask = threading.Lock()    # ... and one unanswered request at a time (reports go on while it waits for its answer)
    try:
        dtrain = Dataset(X, label=y, weight=weight, params=dataset_params, free_raw_data=False)
        dtrain.construct()
        n_started = [0]
        stopped = [False]

        def next_row():
            with ask:
                if stopped[0]:
                    return None
                with send:
                    conn.send(("want",))
                item = conn.recv()
                if item is None:
                    stopped[0] = True
                return item

        def on_start(row_id):
            with send:
                conn.send(("start", row_id))
                if fail_after is not None and n_started[0] >= fail_after:
                    os._exit(17)   # fault injection: die with the row in flight
                n_started[0] += 1

        def on_done(row_id, it, sc, secs):
            with send:
                conn.send(("done", row_id, (it, sc, secs)))

        _drain_rows(next_row, rows, dtrain, cv_kwargs, on_start, on_done, rows_in_flight, batch_rows)
    except BaseException as e:   # report, then die
        try:
            with send:
                conn.send(("error", repr(e)))
        except OSError:
            pass
        raise
    with send:
        conn.send(("exit",))


def grid_search_cv(param_grid, X=None, y=None, dtrain=None, weight=None, nrounds: int = 1000, nfold: int = 5,
                   objective: str = "regression", eval=None, early_stopping_rounds: Optional[int] = 5,
                   showsd: bool = False, stratified: bool = True, seed: int = 0, checkpoint: Optional[str] = None,
                   resume: bool = True, workers: Optional[Sequence[Any]] = None, compat_sentinel: bool = False,
                   dataset_params: Optional[Dict[str, Any]] = None, verbose: bool = False,
                   inject_fail_after: Optional[int] = None, progress: Optional[Callable[[int, int], None]] = None,
                   base_params: Optional[Dict[str, Any]] = None, rows_in_flight: Optional[int] = None,
                   batch_rows: Optional[bool] = None, lockstep_chunk: Optional[int] = None):
    """Runs early-stopped k-fold CV for every row of ``param_grid`` and fills ``iteration`` / ``score``.

    ``workers``: None -> run in this process; otherwise a list of devices, one worker process each,
    e.g. ``[0, 1, ..., 7]`` (GPU ids) or ``["cpu", "cpu"]``.  ``base_params`` are engine parameters every
    row shares (a row's own columns win).  ``rows_in_flight``: grid rows one process evaluates at the same
    time, each from its own host thread (None: 4 on the HIP engine for tables up to 500 k rows, else 1); every
    row is an independent deterministic computation, so the table does not depend on it.  ``batch_rows``: keep
    the rows in flight alive in ONE thread and advance all their fold boosters with single native calls
    (continuous batching; None: on for the HIP build with at most 65536 rows, where a call is one kernel launch
    with one workgroup per booster); ``lockstep_chunk``: iterations per such call (engine.cv).  Returns the grid as a DataFrame (row order unchanged; rank it with
    :func:`rank_grid`).
    """
    import pandas as pd
    grid = param_grid.copy() if hasattr(param_grid, "copy") and hasattr(param_grid, "columns") else pd.DataFrame(list(param_grid))
    for c in RESULT_COLUMNS:
        if c not in grid.columns:
            grid.insert(0 if c == "iteration" else 1, c, np.nan)
    rows = [{**(base_params or {}), **_row_params(r)} for r in grid.to_dict(orient="records")]
    cv_kwargs = dict(nrounds=nrounds, nfold=nfold, objective=objective, eval=eval, showsd=showsd,
                     early_stopping_rounds=early_stopping_rounds, verbose=0, stratified=stratified, seed=seed,
                     compat_sentinel=compat_sentinel, lockstep_chunk=lockstep_chunk)
    extra = {k: cv_kwargs[k] for k in ("nrounds", "nfold", "objective", "eval", "early_stopping_rounds", "seed")}
    hashes = [_params_hash(p, extra) for p in rows]
    ck = GridCheckpoint(checkpoint if (checkpoint and (resume or not os.path.exists(checkpoint))) else checkpoint)
    if checkpoint and not resume and os.path.exists(checkpoint):
        os.remove(checkpoint)
        ck = GridCheckpoint(checkpoint)
    todo = [i for i in range(len(rows)) if not ck.has(i, hashes[i])]
    todo.sort(key=lambda i: -_cost_estimate(rows[i]))     # longest first: better load balance
    total, finished = len(rows), len(rows) - len(todo)

    def record(i, it, sc, secs, worker):
        nonlocal finished
        ck.append({"row": i, "hash": hashes[i], "params": rows[i], "iteration": it, "score": sc, "seconds": secs,
                   "worker": worker, "status": "ok"})
        finished += 1
        if verbose:
            print(f"[grid] row {i + 1}/{total} done ({finished}/{total}): iteration={it} score={sc:.6g} ({secs:.1f}s)")
        if progress:
            progress(finished, total)

    if not workers:
        from .basic import Dataset
        if dtrain is None:
            dtrain = Dataset(X, label=y, weight=weight, params=dataset_params, free_raw_data=False)
        dtrain.construct()
        started = [0]
        pending = iter(todo)

        def on_start(i):
            if inject_fail_after is not None and started[0] >= inject_fail_after:
                raise RuntimeError("injected failure")
            started[0] += 1

        _drain_rows(lambda: next(pending, None), rows, dtrain, cv_kwargs, on_start,
                    lambda i, it, sc, secs: record(i, it, sc, secs, "main"), rows_in_flight, batch_rows)
    else:
        if X is None or y is None:
            raise ValueError("worker processes need the raw X and y (a constructed Dataset cannot cross processes)")
        import multiprocessing as mp
        from multiprocessing.connection import wait as wait_for_any
        ctx = mp.get_context("spawn")
        Xn, yn = np.asarray(X), np.asarray(y)
        procs, conns = {}, {}
        inflight: Dict[int, set] = {}
        for w, dev in enumerate(workers):
            fa = inject_fail_after if w == 0 else None
            here, there = ctx.Pipe(duplex=True)
            p = ctx.Process(target=_worker_main, args=(w, dev, Xn, yn, weight, dataset_params, rows, cv_kwargs, there, fa,
                                                       rows_in_flight, batch_rows), daemon=True)
            p.start()
            there.close()    # the worker holds the only other end: its death shows as end-of-file here
            procs[w], conns[w] = p, here
            inflight[w] = set()
        pending = list(todo)            # rows nobody has been given yet, cheapest last (todo is sorted by cost)
        remaining = set(todo)
        alive = set(procs.keys())

        def lost_worker(w):
            alive.discard(w)
            lost = sorted(r for r in inflight.get(w, ()) if r in remaining)
            if lost:
                if verbose:
                    print(f"[grid] worker {w} died; re-queueing rows {[r + 1 for r in lost]}")
                pending[:0] = lost     # they go first
            inflight[w] = set()

        while remaining and alive:
            ready = wait_for_any([conns[w] for w in alive], timeout=1.0)
            if not ready:
                for w in list(alive):
                    if not procs[w].is_alive() and not conns[w].poll():
                        lost_worker(w)
                continue
            for conn in ready:
                w = next(k for k, c in conns.items() if c is conn)
                try:
                    msg = conn.recv()
                except (EOFError, OSError):
                    lost_worker(w)
                    continue
                kind = msg[0]
                if kind == "want":
                    # answered at once, with "stop" when nothing is pending: a worker that batches its rows asks from
                    # the one thread that also advances them, so it must never be kept waiting.  Rows a worker hands
                    # back by dying go to whoever asks afterwards, else to this process (below).
                    answer = pending.pop(0) if pending else None
                    try:
                        conn.send(answer)
                    except OSError:
                        if answer is not None:
                            pending.insert(0, answer)
                        lost_worker(w)
                elif kind == "start":
                    inflight[w].add(msg[1])
                elif kind == "done":
                    row_id, (it, sc, secs) = msg[1], msg[2]
                    inflight[w].discard(row_id)
                    if row_id in remaining:
                        remaining.discard(row_id)
                        record(row_id, it, sc, secs, str(workers[w]))
                elif kind == "error":
                    if verbose:
                        print(f"[grid] worker {w} failed: {msg[1]}")
                elif kind == "exit":
                    alive.discard(w)
        for w in alive:                 # everything is done: whoever still asks (or is about to) is told to stop
            try:
                conns[w].send(None)
            except OSError:
                pass
        for p in procs.values():
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if remaining:
            # every worker is gone (or left before a re-queued row arrived): finish the rest here
            from .basic import Dataset
            dt = Dataset(Xn, label=yn, weight=weight, params=dataset_params, free_raw_data=False)
            for i in sorted(remaining, key=lambda i: -_cost_estimate(rows[i])):
                t0 = time.time()
                it, sc = _cv_one(rows[i], dt, cv_kwargs)
                record(i, it, sc, time.time() - t0, "main")
    for i in range(len(rows)):
        rec = ck.done.get(i)
        if rec is not None:
            grid.loc[grid.index[i], "iteration"] = rec["iteration"]
            grid.loc[grid.index[i], "score"] = rec["score"]
    return grid


_DIST_CALLS = [0]


def _row_counter(group, token: str):
    """A fetch-and-add counter every rank of ``group`` can reach: a key in the rendezvous store of the default
    process group (TCPStore / FileStore ``add`` is atomic).  None when this build of torch keeps no such store."""
    try:
        import torch.distributed as dist
        store = dist.distributed_c10d._get_default_store()
    except Exception:       # noqa: BLE001 - private accessor; absent means static sharding
        return None
    key = f"lightgbm_amd/grid/{token}/next"
    return lambda: int(store.add(key, 1)) - 1


def grid_search_cv_distributed(param_grid, X, y, group=None, checkpoint: Optional[str] = None, resume: bool = True,
                               weight=None, dataset_params: Optional[Dict[str, Any]] = None,
                               base_params: Optional[Dict[str, Any]] = None, rows_in_flight: Optional[int] = None,
                               verbose: bool = False, static_shards: bool = False, **cv_args):
    """Task-parallel search under ``torch.distributed`` (one rank per GPU): the loop of r/gridsearchCV.R:104-121
    with the rows dealt out dynamically.

    Rank 0 lists the unfinished rows most-expensive-first and broadcasts the list; every rank then pulls the
    next position from a shared counter until the list is exhausted, so a rank that drew short rows simply draws
    more of them.  Each rank appends its finished rows to ``<checkpoint>.rank<r>`` (fsync per row, R:118); a
    restarted search first merges ``<checkpoint>`` and every ``<checkpoint>.rank*`` file, whatever the world
    size of the run that wrote them (R:121).  All ranks return the full table.  ``cv_args`` are those of
    :func:`grid_search_cv` (``nrounds``, ``nfold``, ``objective``, ``eval``, ``early_stopping_rounds``, ...).
    ``static_shards`` forces the fallback used when the process group exposes no store: rank r takes every
    world_size-th row of the cost-sorted list.
    """
    import glob
    import pandas as pd    # noqa: F401
    import torch.distributed as dist
    from .basic import Dataset
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    grid = param_grid.copy()
    for c in RESULT_COLUMNS:
        if c not in grid.columns:
            grid.insert(0 if c == "iteration" else 1, c, np.nan)
    rows = [{**(base_params or {}), **_row_params(r)} for r in grid.to_dict(orient="records")]
    cv_kwargs = dict(nrounds=cv_args.pop("nrounds", 1000), nfold=cv_args.pop("nfold", 5),
                     objective=cv_args.pop("objective", "regression"), eval=cv_args.pop("eval", None),
                     showsd=cv_args.pop("showsd", False), early_stopping_rounds=cv_args.pop("early_stopping_rounds", 5),
                     verbose=0, stratified=cv_args.pop("stratified", True), seed=cv_args.pop("seed", 0),
                     compat_sentinel=cv_args.pop("compat_sentinel", False))
    if cv_args:
        raise TypeError(f"grid_search_cv_distributed: unknown arguments {sorted(cv_args)}")
    extra = {k: cv_kwargs[k] for k in ("nrounds", "nfold", "objective", "eval", "early_stopping_rounds", "seed")}
    hashes = [_params_hash(p, extra) for p in rows]

    # rank 0 decides what is left (one reader of the old files, so every rank works from the same list)
    _DIST_CALLS[0] += 1
    plan: List[Any] = [None]
    if rank == 0:
        done: Dict[int, Dict[str, Any]] = {}
        if checkpoint:
            old = sorted(set(glob.glob(glob.escape(checkpoint) + ".rank*")) | ({checkpoint} if os.path.exists(checkpoint) else set()))
            if not resume:
                for f in old:
                    os.remove(f)
            else:
                for f in old:
                    for i, rec in GridCheckpoint(f).done.items():
                        if i < len(rows) and rec.get("hash") == hashes[i]:
                            done[i] = rec
        todo = [i for i in range(len(rows)) if i not in done]
        todo.sort(key=lambda i: -_cost_estimate(rows[i]))
        token = f"{os.getpid()}-{time.time_ns()}-{_DIST_CALLS[0]}"
        plan = [(todo, {i: (r["iteration"], r["score"]) for i, r in done.items()}, token)]
    dist.broadcast_object_list(plan, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    todo, finished, token = plan[0]

    counter = None if static_shards else _row_counter(group, token)
    if counter is None:
        mine = iter([i for k, i in enumerate(todo) if k % world == rank])
        next_row = lambda: next(mine, None)                       # noqa: E731
    else:
        def next_row():
            k = counter()
            return todo[k] if k < len(todo) else None

    ck = GridCheckpoint(f"{checkpoint}.rank{rank}" if checkpoint else None)
    local: Dict[int, Tuple[float, float]] = {}
    if todo:
        dtrain = Dataset(X, label=y, weight=weight, params=dataset_params, free_raw_data=False)
        dtrain.construct()

        def run_one(i):
            t0 = time.time()
            it, sc = _cv_one(rows[i], dtrain, cv_kwargs)
            ck.append({"row": i, "hash": hashes[i], "params": rows[i], "iteration": it, "score": sc,
                       "seconds": time.time() - t0, "worker": f"rank{rank}", "status": "ok"})
            local[i] = (it, sc)
            if verbose:
                print(f"[grid rank {rank}] row {i + 1}/{len(rows)}: iteration={it} score={sc:.6g} ({time.time() - t0:.1f}s)")

        _run_rows(next_row, run_one, rows_in_flight or _auto_rows_in_flight(dtrain, rows))
    gathered: List[Any] = [None] * world
    dist.all_gather_object(gathered, local, group=group)
    for part in [finished] + gathered:
        for i, (it, sc) in part.items():
            grid.loc[grid.index[int(i)], "iteration"] = it
            grid.loc[grid.index[int(i)], "score"] = sc
    return grid


def rank_grid(grid):
    """Best first: descending ``score`` (scores are negated losses); unfinished / sentinel rows last."""
    g = grid.copy()
    key = g["score"].where(g["iteration"] > 0, other=-np.inf).fillna(-np.inf)
    return g.assign(_k=key).sort_values("_k", ascending=False, kind="mergesort").drop(columns="_k")


def average_predictions(pred_list: Sequence[np.ndarray]) -> np.ndarray:
    """Unweighted mean of prediction vectors (``rowMeans(do.call(cbind, predList))``, R:143)."""
    return np.mean(np.column_stack([np.asarray(p, dtype=np.float64) for p in pred_list]), axis=1)


def refit_top(ranked_grid, m: int, dtrain, X_test, objective: str = "regression", keep_models: bool = False,
              **train_kwargs):
    """Refits the best ``m`` rows with their CV-chosen number of rounds; returns (mean prediction, per-model list).

    Rows whose ``iteration`` is not a positive number (never early-stopped under ``compat_sentinel``) are refused:
    passing -1 as the number of rounds is the latent hazard of R:134-138.
    """
    from .rstyle import lgb_train
    preds, models = [], []
    top = ranked_grid.head(m)
    for _, row in top.iterrows():
        it = row["iteration"]
        if not (it == it) or it < 1:
            raise ValueError("refit_top: a selected grid row has no valid iteration count")
        params = _row_params(row.to_dict())
        fit = lgb_train(params, dtrain, nrounds=int(it), objective=objective, verbose=0, **train_kwargs)
        preds.append(fit.predict(X_test, num_iteration=-1))
        if keep_models:
            models.append(fit)
    mean = average_predictions(preds)
    return (mean, preds, models) if keep_models else (mean, preds)
