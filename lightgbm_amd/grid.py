"""Parameter-grid cross-validation search: the headline workload of the reference.

Mirrors r/gridsearchCV.R:92-144:

* :func:`expand_grid`       <- ``expand.grid(...)`` (R:92-100): Cartesian product, the FIRST
  factor varies fastest, result columns ``iteration`` / ``score`` pre-allocated as NaN.
* :func:`grid_search_cv`    <- the ``for`` loop over grid rows running ``lgb.cv`` (R:104-119),
  re-organised as a work queue: rows are ordered most-expensive-first and pulled by one worker
  process per GPU (or CPU worker); every finished row is appended to a durable JSON-lines
  checkpoint (fsync per row), the analogue of ``save(paramGrid, ...)`` after every row (R:118);
  a restarted search skips finished rows (R:121 ``load("paramGrid.RData")``); a worker that dies
  hands its in-flight row back to the queue.
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
        self.path = path
        self.done: Dict[int, Dict[str, Any]] = {}
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


def _worker_main(worker_id: int, device: Optional[str], X, y, weight, dataset_params, rows, cv_kwargs, task_q, result_q,
                 fail_after: Optional[int]):
    """Worker process: pulls row ids until it sees None."""
    if device is not None and device != "cpu":
        os.environ["HIP_VISIBLE_DEVICES"] = str(device)
        os.environ["CUDA_VISIBLE_DEVICES"] = str(device)
    elif device == "cpu":
        os.environ["LGA_FORCE_HOST"] = "1"
    from .basic import Dataset
    try:
        dtrain = Dataset(X, label=y, weight=weight, params=dataset_params, free_raw_data=False)
        dtrain.construct()
        n_done = 0
        while True:
            item = task_q.get()
            if item is None:
                break
            row_id = item
            result_q.put(("start", worker_id, row_id, None))
            t0 = time.time()
            if fail_after is not None and n_done >= fail_after:
                os._exit(17)   # fault injection: die with the row in flight
            it, sc = _cv_one(rows[row_id], dtrain, cv_kwargs)
            result_q.put(("done", worker_id, row_id, (it, sc, time.time() - t0)))
            n_done += 1
    except BaseException as e:   # report, then die
        result_q.put(("error", worker_id, -1, repr(e)))
        raise
    result_q.put(("exit", worker_id, -1, None))


def grid_search_cv(param_grid, X=None, y=None, dtrain=None, weight=None, nrounds: int = 1000, nfold: int = 5,
                   objective: str = "regression", eval=None, early_stopping_rounds: Optional[int] = 5,
                   showsd: bool = False, stratified: bool = True, seed: int = 0, checkpoint: Optional[str] = None,
                   resume: bool = True, workers: Optional[Sequence[Any]] = None, compat_sentinel: bool = False,
                   dataset_params: Optional[Dict[str, Any]] = None, verbose: bool = False,
                   inject_fail_after: Optional[int] = None, progress: Optional[Callable[[int, int], None]] = None,
                   base_params: Optional[Dict[str, Any]] = None):
    """Runs early-stopped k-fold CV for every row of ``param_grid`` and fills ``iteration`` / ``score``.

    ``workers``: None -> run in this process; otherwise a list of devices, one worker process each,
    e.g. ``[0, 1, ..., 7]`` (GPU ids) or ``["cpu", "cpu"]``.  ``base_params`` are engine parameters every
    row shares (a row's own columns win).  Returns the grid as a DataFrame (row order unchanged; rank it
    with :func:`rank_grid`).
    """
    import pandas as pd
    grid = param_grid.copy() if hasattr(param_grid, "copy") and hasattr(param_grid, "columns") else pd.DataFrame(list(param_grid))
    for c in RESULT_COLUMNS:
        if c not in grid.columns:
            grid.insert(0 if c == "iteration" else 1, c, np.nan)
    rows = [{**(base_params or {}), **_row_params(r)} for r in grid.to_dict(orient="records")]
    cv_kwargs = dict(nrounds=nrounds, nfold=nfold, objective=objective, eval=eval, showsd=showsd,
                     early_stopping_rounds=early_stopping_rounds, verbose=0, stratified=stratified, seed=seed,
                     compat_sentinel=compat_sentinel)
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
        done_here = 0
        for i in todo:
            if inject_fail_after is not None and done_here >= inject_fail_after:
                raise RuntimeError("injected failure")
            t0 = time.time()
            it, sc = _cv_one(rows[i], dtrain, cv_kwargs)
            record(i, it, sc, time.time() - t0, "main")
            done_here += 1
    else:
        if X is None or y is None:
            raise ValueError("worker processes need the raw X and y (a constructed Dataset cannot cross processes)")
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        task_q, result_q = ctx.Queue(), ctx.Queue()
        for i in todo:
            task_q.put(i)
        Xn, yn = np.asarray(X), np.asarray(y)
        procs = {}
        inflight: Dict[int, Optional[int]] = {}
        for w, dev in enumerate(workers):
            fa = inject_fail_after if w == 0 else None
            p = ctx.Process(target=_worker_main, args=(w, dev, Xn, yn, weight, dataset_params, rows, cv_kwargs, task_q,
                                                       result_q, fa), daemon=True)
            p.start()
            procs[w] = p
            inflight[w] = None
        for _ in workers:
            task_q.put(None)
        remaining = set(todo)
        alive = set(procs.keys())
        import queue as _q
        while remaining and alive:
            try:
                kind, w, row_id, payload = result_q.get(timeout=1.0)
            except _q.Empty:
                for w in list(alive):
                    if not procs[w].is_alive():
                        alive.discard(w)
                        lost = inflight.get(w)
                        if lost is not None and lost in remaining:
                            if verbose:
                                print(f"[grid] worker {w} died; re-queueing row {lost + 1}")
                            # the dead worker never consumed its stop marker: the re-queued row goes in front of it
                            task_q.put(lost)
                            task_q.put(None)
                            inflight[w] = None
                continue
            if kind == "start":
                inflight[w] = row_id
            elif kind == "done":
                it, sc, secs = payload
                inflight[w] = None
                if row_id in remaining:
                    remaining.discard(row_id)
                    record(row_id, it, sc, secs, str(workers[w]))
            elif kind == "error":
                if verbose:
                    print(f"[grid] worker {w} failed: {payload}")
            elif kind == "exit":
                alive.discard(w)
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


def grid_search_cv_distributed(param_grid, X, y, group=None, **kwargs):
    """Task-parallel search under ``torch.distributed``: rank r evaluates every world_size-th row of the
    cost-sorted grid on its own GPU; results are all-gathered so every rank returns the full table."""
    import pandas as pd
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    grid = param_grid.copy()
    for c in RESULT_COLUMNS:
        if c not in grid.columns:
            grid.insert(0 if c == "iteration" else 1, c, np.nan)
    rows = [_row_params(r) for r in grid.to_dict(orient="records")]
    order = sorted(range(len(rows)), key=lambda i: -_cost_estimate(rows[i]))
    mine = [i for k, i in enumerate(order) if k % world == rank]
    sub = grid.iloc[mine].copy()
    ck = kwargs.pop("checkpoint", None)
    if ck:
        ck = f"{ck}.rank{rank}"
    res = grid_search_cv(sub, X=X, y=y, checkpoint=ck, workers=None, **kwargs)
    local = [(int(i), float(res.iloc[k]["iteration"]), float(res.iloc[k]["score"])) for k, i in enumerate(mine)]
    gathered: List[Any] = [None] * world
    dist.all_gather_object(gathered, local, group=group)
    for part in gathered:
        for i, it, sc in part:
            grid.loc[grid.index[i], "iteration"] = it
            grid.loc[grid.index[i], "score"] = sc
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
