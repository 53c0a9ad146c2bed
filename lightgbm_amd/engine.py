"""``train`` and ``cv``: the two training entry points the reference uses.

* ``train``  <- ``lgb.train(params, data, nrounds, objective=, verbose=)``
  (r/gridsearchCV.R:57-61, 134-138)
* ``cv``     <- ``lgb.cv(params, data, nrounds, nfold, objective=, eval=, showsd=,
  early_stopping_rounds=, verbose=)`` (r/gridsearchCV.R:70-78, 108-115): k fold
  boosters share ONE binned dataset (folds are row-index subsets), advance in
  lock-step, and early stopping watches the mean of the fold metrics.
"""
from __future__ import annotations

import math
import json

import copy
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np

from . import callback as cb
from .basic import Booster, Dataset, LightGBMError, get_param, pop_param, resolve_alias

__all__ = ["train", "cv", "CVBooster", "make_folds"]


def _split_callbacks(callbacks):
    cbs = set(callbacks or [])
    before = {c for c in cbs if getattr(c, "before_iteration", False)}
    after = cbs - before
    key = lambda c: getattr(c, "order", 0)
    return sorted(before, key=key), sorted(after, key=key)


def _prepare_params(params: Optional[Dict[str, Any]], kwargs: Dict[str, Any]) -> Dict[str, Any]:
    """params + extra named arguments share one namespace (the R binding's `...`, r/gridsearchCV.R:60)."""
    p = copy.deepcopy(dict(params or {}))
    for k, v in kwargs.items():
        if resolve_alias(k) is not None:
            p[k] = v
        else:
            raise TypeError(f"unexpected keyword argument {k!r}")
    return p


def train(params: Optional[Dict[str, Any]], train_set: Dataset, num_boost_round: int = 100,
          valid_sets: Optional[List[Dataset]] = None, valid_names: Optional[List[str]] = None,
          fobj: Optional[Callable] = None, feval=None, init_model: Union[str, Booster, None] = None,
          early_stopping_rounds: Optional[int] = None, evals_result: Optional[dict] = None,
          verbose_eval: Union[bool, int] = False, keep_training_booster: bool = False,
          callbacks: Optional[List[Callable]] = None, comm=None, **kwargs) -> Booster:
    """Trains a booster for ``num_boost_round`` iterations (fewer if stopped early)."""
    params = _prepare_params(params, kwargs)
    nit = pop_param(params, "num_iterations")
    if nit is not None:
        num_boost_round = int(nit)
    esr = pop_param(params, "early_stopping_round")
    if esr is not None and early_stopping_rounds is None:
        early_stopping_rounds = int(esr)
    first_metric_only = bool(get_param(params, "first_metric_only", False))
    if num_boost_round <= 0:
        raise ValueError("num_boost_round should be greater than zero.")
    if callable(params.get("objective")):
        fobj = params.pop("objective")
    if fobj is not None:
        params["objective"] = "custom"

    predictor: Optional[Booster] = None
    if isinstance(init_model, str):
        predictor = Booster(model_file=init_model)
    elif isinstance(init_model, Booster):
        predictor = init_model
    if predictor is not None:
        # continue training: the old model's raw predictions become the data's initial scores
        raw = train_set.data
        if raw is None:
            raise LightGBMError("init_model needs the raw training data (construct the Dataset with free_raw_data=False)")
        train_set.set_init_score(predictor.predict(raw, raw_score=True, num_iteration=-1))

    booster = Booster(params=params, train_set=train_set, comm=comm)
    is_valid_contain_train = False
    train_data_name = "training"
    if valid_sets is not None:
        if isinstance(valid_sets, Dataset):
            valid_sets = [valid_sets]
        if isinstance(valid_names, str):
            valid_names = [valid_names]
        for i, vd in enumerate(valid_sets):
            name = valid_names[i] if valid_names is not None and len(valid_names) > i else f"valid_{i}"
            if vd is train_set:
                is_valid_contain_train = True
                train_data_name = name if valid_names is not None and len(valid_names) > i else "training"
                continue
            if predictor is not None and vd.data is not None:
                vd.set_init_score(predictor.predict(vd.data, raw_score=True, num_iteration=-1))
            booster.add_valid(vd, name)
    booster._train_data_name = train_data_name

    callbacks = list(callbacks or [])
    if verbose_eval is True:
        callbacks.append(cb.log_evaluation())
    elif isinstance(verbose_eval, int) and not isinstance(verbose_eval, bool) and verbose_eval > 0:
        callbacks.append(cb.log_evaluation(verbose_eval))
    if early_stopping_rounds is not None and early_stopping_rounds > 0:
        callbacks.append(cb.early_stopping(early_stopping_rounds, first_metric_only, verbose=bool(verbose_eval),
                                           min_delta=float(get_param(params, "early_stopping_min_delta", 0.0) or 0.0)))
    if evals_result is not None:
        callbacks.append(cb.record_evaluation(evals_result))
    cb_before, cb_after = _split_callbacks(callbacks)

    # periodic snapshots (snapshot_freq / save_period): "<output_model>.snapshot_iter_<k>", written atomically, so a
    # killed run can continue with train(init_model=<latest snapshot>)
    snapshot_freq = int(get_param(params, "snapshot_freq", -1) or -1)
    snapshot_base = str(get_param(params, "output_model", "LightGBM_model.txt"))

    need_eval = bool(booster.valid_sets) or is_valid_contain_train
    booster.best_iteration = 0
    evaluation_result_list: List[Tuple] = []
    # Small data on the HIP engine CAN run whole chunks of iterations on the device without Python - or the host - in
    # between (Booster.update_group); the callbacks are replayed over a chunk's evaluation results and iterations
    # trained beyond an early stop are rolled back, so the model is the plain loop's.  That path is one workgroup per
    # booster - made for many boosters at once (cv, grid search); a booster that trains alone is faster with growth
    # rounds over the whole device, so train() takes it only when whole_tree_max_rows is given explicitly.
    from . import _lib
    chunk = 0
    if (get_param(params, "whole_tree_max_rows") is not None and int(get_param(params, "whole_tree_max_rows")) > 0 and
            _lib.is_hip() and str(get_param(params, "device_type", "gpu")).lower() not in ("cpu", "host") and comm is None
            and fobj is None and feval is None and not cb_before and snapshot_freq <= 0 and not is_valid_contain_train
            and len(booster.valid_sets) <= 1 and train_set.num_data() <= 65536
            and str(get_param(params, "boosting", "gbdt")).lower() != "dart"):     # (a dart iteration cannot be rolled back)
        chunk = 10 if need_eval else 50
    i = 0
    while chunk and i < num_boost_round:
        values, no_split, done = Booster.update_group([booster], min(chunk, num_boost_round - i), need_eval)
        names = booster._metric_names() if need_eval else []
        stop = False
        for j in range(done):
            evaluation_result_list = [(booster.name_valid_sets[0], mn, float(values[j, 0, m]), hb) for m, (mn, hb) in enumerate(names)]
            try:
                for c in cb_after:
                    c(cb.CallbackEnv(booster, params, i + j, 0, num_boost_round, evaluation_result_list))
            except cb.EarlyStopException as es:
                booster.best_iteration = es.best_iteration + 1
                evaluation_result_list = es.best_score
                stop = True
            if stop or bool(no_split[j, 0]):
                for _ in range(int((~no_split[j + 1: done, 0]).sum())):
                    booster.rollback_one_iter()
                stop = True
                break
        i += done
        if stop:
            break
    for i in range(num_boost_round if not chunk else 0):
        for c in cb_before:
            c(cb.CallbackEnv(booster, params, i, 0, num_boost_round, None))
        finished = booster.update(fobj=fobj)
        if snapshot_freq > 0 and (i + 1) % snapshot_freq == 0:
            booster.save_model(f"{snapshot_base}.snapshot_iter_{i + 1}")
        evaluation_result_list = []
        if need_eval:
            if is_valid_contain_train:
                evaluation_result_list.extend(booster.eval_train(feval))
            evaluation_result_list.extend(booster.eval_valid(feval))
        try:
            for c in cb_after:
                c(cb.CallbackEnv(booster, params, i, 0, num_boost_round, evaluation_result_list))
        except cb.EarlyStopException as es:
            booster.best_iteration = es.best_iteration + 1
            evaluation_result_list = es.best_score
            break
        if finished:
            break
    booster.best_score = {}
    for item in evaluation_result_list or []:
        booster.best_score.setdefault(item[0], {})[item[1]] = item[2]
    if predictor is not None:
        booster._prepend(predictor)
        if booster.best_iteration > 0:
            booster.best_iteration += predictor.current_iteration()
    if not keep_training_booster:
        booster = _carry(booster)
    return booster


def _carry(trained: Booster) -> Booster:
    """Turns a training booster into a light prediction booster, keeping its bookkeeping."""
    out = Booster(model_str=trained.model_to_string(num_iteration=-1))
    out.best_iteration = trained.best_iteration
    out.best_score = trained.best_score
    out.params = trained.params
    return out



# ----------------------------------------------------------------------------
# cross-validation
# ----------------------------------------------------------------------------
class CVBooster:
    """The k fold boosters of one ``cv`` run.

    ``best_iteration`` is 1-based like upstream's Python API; the R-style
    accessors used by the reference (``best_iter``, ``best_score`` with flipped
    sign for losses) are provided by :mod:`lightgbm_amd.rstyle`.
    """

    def __init__(self):
        self.boosters: List[Booster] = []
        self.best_iteration = -1
        self.iterations_run = 0
        self.higher_better: Dict[str, bool] = {}   # per recorded metric: is a larger value the better one?

    def append(self, booster: Booster):
        self.boosters.append(booster)

    def __getattr__(self, name):
        def handler(*args, **kwargs):
            return [getattr(b, name)(*args, **kwargs) for b in self.boosters]
        return handler


def _stratify_labels(y: np.ndarray, nfold: int, classification: bool) -> np.ndarray:
    """Group id per row to stratify on: the class, or (numeric labels) a quantile group, caret style."""
    if classification:
        _, inv = np.unique(y, return_inverse=True)
        return inv
    n = len(y)
    groups = min(5, max(2, n // nfold))
    qs = np.quantile(y, np.linspace(0, 1, groups + 1)[1:-1])
    return np.searchsorted(np.unique(qs), y, side="right")


def make_folds(n: int, nfold: int, label: Optional[np.ndarray] = None, stratified: bool = True,
               shuffle: bool = True, seed: int = 0, classification: bool = False) -> List[Tuple[np.ndarray, np.ndarray]]:
    """(train_idx, test_idx) pairs: disjoint test folds that cover 0..n-1, sizes differing by at most one per stratum."""
    if nfold < 2:
        raise ValueError("nfold must be at least 2")
    if n < nfold:
        raise ValueError("more folds than rows")
    rng = np.random.RandomState(seed)
    fold_of = np.empty(n, dtype=np.int64)
    if stratified and label is not None:
        strata = _stratify_labels(np.asarray(label), nfold, classification)
        offset = 0
        for g in np.unique(strata):
            idx = np.flatnonzero(strata == g)
            if shuffle:
                rng.shuffle(idx)
            # continue dealing where the previous stratum stopped so fold sizes stay balanced
            fold_of[idx] = (np.arange(len(idx)) + offset) % nfold
            offset = (offset + len(idx)) % nfold
    else:
        idx = np.arange(n)
        if shuffle:
            rng.shuffle(idx)
        bounds = np.linspace(0, n, nfold + 1).astype(np.int64)
        for k in range(nfold):
            fold_of[idx[bounds[k]:bounds[k + 1]]] = k
    all_idx = np.arange(n)
    return [(all_idx[fold_of != k], all_idx[fold_of == k]) for k in range(nfold)]


class _FoldRunner:
    """Runs the k fold boosters of a CV either one after the other or - HIP build, small data - concurrently:
    fold i owns host thread i and HIP stream i, so the latency-bound growth rounds of small folds
    (45 k rows in the reference, r/gridsearchCV.R:26-27) overlap on the GPU instead of queueing.
    Results do not depend on the mode: every fold is an independent, deterministic computation."""

    def __init__(self, k: int, concurrent: bool):
        self.k = k
        self.concurrent = concurrent and k > 1
        if self.concurrent:
            import torch
            from concurrent.futures import ThreadPoolExecutor
            # the shared binned dataset was built on the caller's stream: complete before side streams read it
            # (not a device-wide wait - other grid rows may be running on their own streams)
            torch.cuda.current_stream().synchronize()
            self.streams = [torch.cuda.Stream() for _ in range(k)]
            self.pools = [ThreadPoolExecutor(max_workers=1) for _ in range(k)]

    def _enter(self, i):
        import contextlib
        if not self.concurrent:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.streams[i])

    def construct(self, i, fn):
        """Boosters are created from the calling thread (they share the root dataset), but on fold i's stream."""
        with self._enter(i):
            return fn()

    def map(self, fns):
        """fns[i]() for every fold; concurrently when enabled."""
        if not self.concurrent:
            return [fn() for fn in fns]

        def run(i, fn):
            from . import _device, _lib
            with self._enter(i):
                _device.sync_stream(_lib.load()[0])
                return fn()
        futures = [self.pools[i].submit(run, i, fn) for i, fn in enumerate(fns)]
        return [f.result() for f in futures]

    def close(self):
        if self.concurrent:
            for s in self.streams:
                s.synchronize()
            for p in self.pools:
                p.shutdown(wait=True)


def _agg_cv_result(raw_results: List[List[Tuple]], eval_train_metric: bool):
    """Mean / stdv over folds of every (data, metric) pair, in first-seen order."""
    cvmap: Dict[str, List[float]] = {}
    metric_type: Dict[str, bool] = {}
    order: List[str] = []
    for one_result in raw_results:
        for one_line in one_result:
            key = f"{one_line[0]} {one_line[1]}" if eval_train_metric else one_line[1]
            if key not in cvmap:
                order.append(key)
                cvmap[key] = []
            metric_type[key] = one_line[3]
            cvmap[key].append(one_line[2])
    out = []
    for k in order:
        mean, std = _mean_std(cvmap[k])
        out.append(("cv_agg", k, mean, metric_type[k], std))
    return out


def _mean_std(v: List[float]) -> Tuple[float, float]:
    """(numpy.mean(v), numpy.std(v)) of a handful of fold values by the operations numpy performs on so few elements - a
    running sum in index order, squared deviations by multiplication - without its per-call overhead (a grid search
    aggregates tens of thousands of five-element lists); longer lists go to numpy, whose pairwise summation they would
    take there."""
    n = len(v)
    if n == 0 or n >= 8:
        return float(np.mean(v)), float(np.std(v))
    total = 0.0
    for x in v:
        total += x
    mean = total / n
    sq = 0.0
    for x in v:
        d = x - mean
        sq += d * d
    return mean, math.sqrt(sq / n)


def cv(params: Optional[Dict[str, Any]], train_set: Dataset, num_boost_round: int = 100, folds=None, nfold: int = 5,
       stratified: bool = True, shuffle: bool = True, metrics: Union[str, List[str], None] = None,
       fobj: Optional[Callable] = None, feval=None, init_model=None, early_stopping_rounds: Optional[int] = None,
       fpreproc=None, verbose_eval: Union[bool, int, None] = None, show_stdv: bool = True, seed: int = 0,
       callbacks: Optional[List[Callable]] = None, eval_train_metric: bool = False, return_cvbooster: bool = False,
       concurrent_folds: Optional[bool] = None, lockstep_chunk: Optional[int] = None, **kwargs) -> Dict[str, Any]:
    """k-fold cross-validation; returns ``{"<metric>-mean": [...], "<metric>-stdv": [...]}`` (+ ``"cvbooster"``).

    ``concurrent_folds``: drive the fold boosters from k host threads on k HIP streams (default: on for the
    HIP build when the data has at most 500 k rows - larger folds fill the GPU on their own - and no
    Python-side objective / metric is used).

    ``lockstep_chunk``: run the folds in lockstep inside the native library, this many iterations per call
    (:meth:`Booster.update_group`): on the HIP engine small folds then run whole chunks of iterations on the device
    in one kernel launch, and Python sees the evaluation results once per chunk.  The callbacks are replayed over
    the chunk's results iteration by iteration and iterations trained beyond an early stop are rolled back, so the
    outcome is that of the plain loop.  Default: on (10 iterations per chunk) for the HIP
    build with at most 65536 training rows per fold when nothing needs Python between iterations (no ``fobj`` /
    ``feval`` / ``fpreproc``, no training-data metric, only evaluation callbacks); 0 switches it off."""
    return drive_lockstep(cv_steps(params, train_set, num_boost_round, folds, nfold, stratified, shuffle, metrics, fobj, feval,
                                   init_model, early_stopping_rounds, fpreproc, verbose_eval, show_stdv, seed, callbacks,
                                   eval_train_metric, return_cvbooster, concurrent_folds, lockstep_chunk, **kwargs))


def drive_lockstep(steps):
    """Runs a generator of lockstep requests - ``(boosters, n_iters)``, answered with
    :meth:`Booster.update_group`'s result - to its end and returns its value.  Schedulers that batch the
    requests of MANY such generators into one native call (the grid search) drive the generators themselves."""
    try:
        request = next(steps)
        while True:
            boosters, n_iters = request
            request = steps.send(Booster.update_group(boosters, n_iters, True))
    except StopIteration as stop:
        return stop.value


def cv_steps(params: Optional[Dict[str, Any]], train_set: Dataset, num_boost_round: int = 100, folds=None, nfold: int = 5,
       stratified: bool = True, shuffle: bool = True, metrics: Union[str, List[str], None] = None,
       fobj: Optional[Callable] = None, feval=None, init_model=None, early_stopping_rounds: Optional[int] = None,
       fpreproc=None, verbose_eval: Union[bool, int, None] = None, show_stdv: bool = True, seed: int = 0,
       callbacks: Optional[List[Callable]] = None, eval_train_metric: bool = False, return_cvbooster: bool = False,
       concurrent_folds: Optional[bool] = None, lockstep_chunk: Optional[int] = None, rollback_overrun: bool = True,
       **kwargs):
    """:func:`cv` as a generator: in lockstep mode it yields ``(fold boosters, n_iters)`` whenever it wants the
    folds advanced and expects :meth:`Booster.update_group`'s result back; its return value is :func:`cv`'s.
    ``rollback_overrun=False`` leaves the iterations a chunk trained beyond an early stop in the fold boosters
    (callers that only read the evaluation history)."""
    if not isinstance(train_set, Dataset):
        raise TypeError("Training only accepts Dataset object")
    params = _prepare_params(params, kwargs)
    nit = pop_param(params, "num_iterations")
    if nit is not None:
        num_boost_round = int(nit)
    esr = pop_param(params, "early_stopping_round")
    if esr is not None and early_stopping_rounds is None:
        early_stopping_rounds = int(esr)
    first_metric_only = bool(get_param(params, "first_metric_only", False))
    if num_boost_round <= 0:
        raise ValueError("num_boost_round should be greater than zero.")
    if callable(params.get("objective")):
        fobj = params.pop("objective")
    if fobj is not None:
        params["objective"] = "custom"
    if metrics is not None:
        pop_param(params, "metric")
        params["metric"] = metrics
    if init_model is not None:
        raise LightGBMError("cv(init_model=...) is not supported")

    train_set.params = {**params, **train_set.params} if train_set._handle is None else train_set.params
    train_set.construct()
    n = train_set.num_data()
    objective = str(get_param(params, "objective", "regression"))
    classification = objective in ("binary", "binary_logloss", "binary:logistic", "multiclass", "softmax",
                                   "multiclassova", "multiclass_ova", "ova", "ovr")
    group = train_set.get_group()
    qid = np.repeat(np.arange(len(group)), group) if group is not None else None
    drawn = None
    if folds is not None:
        if hasattr(folds, "split"):
            y = train_set.get_label()
            folds = list(folds.split(X=np.empty((n, 1)), y=y, groups=qid))
        else:
            folds = [(np.asarray(a), np.asarray(b)) for a, b in folds]
    elif qid is not None:
        # ranking data: whole queries are dealt to the folds
        if len(group) < nfold:
            raise ValueError("more folds than queries")
        folds = [(np.flatnonzero(np.isin(qid, tr_q)), np.flatnonzero(np.isin(qid, te_q)))
                 for tr_q, te_q in make_folds(len(group), nfold, None, False, shuffle, seed)]
    else:
        # the same folds are asked for again and again (every row of a grid search draws them with the same seed): a
        # table keeps the last few draws, next to the label array they were drawn for
        key = (n, nfold, bool(stratified), bool(shuffle), int(seed), bool(classification))
        label = train_set.get_label() if stratified else None
        cache = train_set.__dict__.setdefault("_fold_draws", {})
        hit = cache.get(key)
        if hit is not None and (label is None or (hit[0] is not None and np.array_equal(hit[0], label))):
            folds = hit[1]
        else:
            folds = make_folds(n, nfold, label, stratified, shuffle, seed, classification)
            if len(cache) >= 4:
                cache.clear()
            cache[key] = hit = [None if label is None else np.array(label, copy=True), folds, None]
        drawn = hit   # [label the folds were drawn for, folds, digest of the folds once somebody has needed it]

    from . import _lib
    on_gpu = _lib.is_hip() and str(get_param(params, "device_type", "gpu")).lower() not in ("cpu", "host")
    if concurrent_folds is None:
        concurrent_folds = on_gpu and n <= 500_000 and fobj is None and feval is None
    callbacks = list(callbacks or [])
    if lockstep_chunk is None:
        lockstep_chunk = 0
        plain_callbacks = not any(getattr(c, "before_iteration", False) for c in callbacks)
        if (on_gpu and fobj is None and feval is None and fpreproc is None and not eval_train_metric and plain_callbacks
                and max(len(tr) for tr, _ in folds) <= 65536 and int(get_param(params, "num_class", 1) or 1) <= 1
                and str(get_param(params, "boosting", "gbdt")).lower() not in ("dart",)):
            lockstep_chunk = 10
    # an explicit lockstep_chunk gives way where the chunked loop could not reproduce the plain one: iterations of a
    # dart model cannot be rolled back after an early stop, and callbacks that act before an iteration would fire
    # once per chunk
    lockstep = (int(lockstep_chunk or 0) > 0 and fobj is None and feval is None and not eval_train_metric
                and str(get_param(params, "boosting", "gbdt")).lower() != "dart"
                and not any(getattr(c, "before_iteration", False) for c in callbacks))
    runner = _FoldRunner(len(folds), bool(concurrent_folds) and on_gpu and not lockstep)

    # The same folds come back for every row of a parameter grid: their (train, test) subsets - index lists over
    # the shared bins - are kept on the dataset and reused while the dataset parameters agree.
    fold_sets = None
    if fpreproc is None:
        digest = drawn[2] if drawn is not None else None
        if digest is None:
            import hashlib
            h = hashlib.sha1()
            for tr_idx, te_idx in folds:
                h.update(np.ascontiguousarray(te_idx, dtype=np.int64).tobytes())
                h.update(b"|")
            digest = h.hexdigest()
            if drawn is not None:
                drawn[2] = digest
        key = (digest, len(folds), json.dumps(train_set.params, sort_keys=True, default=str))
        cache = train_set.__dict__.setdefault("_cv_fold_sets", {})
        fold_sets = cache.get(key)
        if fold_sets is None:
            if len(cache) >= 4:
                cache.clear()
            fold_sets = cache[key] = [(train_set.subset(tr_idx).construct(), train_set.subset(te_idx).construct())
                                      for tr_idx, te_idx in folds]

    cvbooster = CVBooster()
    for k, (tr_idx, te_idx) in enumerate(folds):
        def build(tr_idx=tr_idx, te_idx=te_idx, k=k):
            if fold_sets is not None:
                tr, te = fold_sets[k]
            else:
                tr = train_set.subset(tr_idx)
                te = train_set.subset(te_idx)
            fold_params = params
            if fpreproc is not None:
                tr, te, fold_params = fpreproc(tr, te, dict(params))
            b = Booster(fold_params, tr)
            if eval_train_metric:
                b._train_data_name = "train"
            b.add_valid(te, "valid")
            return b
        cvbooster.append(runner.construct(k, build))

    if early_stopping_rounds is not None and early_stopping_rounds > 0:
        callbacks.append(cb.early_stopping(early_stopping_rounds, first_metric_only, verbose=False,
                                           min_delta=float(get_param(params, "early_stopping_min_delta", 0.0) or 0.0)))
    if verbose_eval is True:
        callbacks.append(cb.log_evaluation(show_stdv=show_stdv))
    elif isinstance(verbose_eval, int) and not isinstance(verbose_eval, bool) and verbose_eval and verbose_eval > 0:
        callbacks.append(cb.log_evaluation(verbose_eval, show_stdv=show_stdv))
    cb_before, cb_after = _split_callbacks(callbacks)

    results: Dict[str, List[float]] = {}

    def after_iteration(i, raw) -> bool:
        """Aggregates iteration i's fold results and runs the callbacks; True = early stop."""
        cvbooster.iterations_run = i + 1
        res = _agg_cv_result(raw, eval_train_metric)
        for _, key, mean, higher_better, std in res:
            results.setdefault(f"{key}-mean", []).append(mean)
            results.setdefault(f"{key}-stdv", []).append(std)
            cvbooster.higher_better[key] = bool(higher_better)
        try:
            for c in cb_after:
                c(cb.CallbackEnv(cvbooster, params, i, 0, num_boost_round, res))
        except cb.EarlyStopException as es:
            cvbooster.best_iteration = es.best_iteration + 1
            for k in results:
                results[k] = results[k][: cvbooster.best_iteration]
            return True
        return False

    try:
        if lockstep:
            names = cvbooster.boosters[0]._metric_names()
            valid_name = cvbooster.boosters[0].name_valid_sets[0]
            i = 0
            stop = False
            while i < num_boost_round and not stop:
                for c in cb_before:
                    c(cb.CallbackEnv(cvbooster, params, i, 0, num_boost_round, None))
                n_it = min(int(lockstep_chunk), num_boost_round - i)
                # the answer may hold fewer iterations than asked for (a scheduler batching several searches)
                values, no_split, done = yield (cvbooster.boosters, n_it)
                for j in range(done):
                    raw = [[(valid_name, mn, float(values[j, b, m]), hb) for m, (mn, hb) in enumerate(names)]
                           for b in range(len(cvbooster.boosters))]
                    if after_iteration(i + j, raw) or bool(no_split[j].all()):
                        # iterations of this chunk that the plain loop would never have run
                        for b, bst in enumerate(cvbooster.boosters if rollback_overrun else []):
                            for _ in range(int((~no_split[j + 1: done, b]).sum())):
                                bst.rollback_one_iter()
                        stop = True
                        break
                i += done
        else:
            for i in range(num_boost_round):
                for c in cb_before:
                    c(cb.CallbackEnv(cvbooster, params, i, 0, num_boost_round, None))

                def step(b):
                    done = b.update(fobj=fobj)
                    one = []
                    if eval_train_metric:
                        one.extend(b.eval_train(feval))
                    one.extend(b.eval_valid(feval))
                    return done, one
                stepped = runner.map([(lambda b=b: step(b)) for b in cvbooster.boosters])
                finished = [d for d, _ in stepped]
                if after_iteration(i, [one for _, one in stepped]):
                    break
                if all(finished):
                    break
    finally:
        runner.close()
    out: Dict[str, Any] = dict(results)
    if return_cvbooster:
        out["cvbooster"] = cvbooster
    return out
