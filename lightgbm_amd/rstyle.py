"""R-binding flavoured entry points, so r/gridsearchCV.R translates line by line.

================================  =============================================
R (reference)                      here
================================  =============================================
``lgb.Dataset(X, label = y)``       ``lgb_Dataset(X, label=y)``           (R:52-53)
``lgb.train(params, data, nrounds,  ``lgb_train(params, data, nrounds,
objective =, verbose =)``           objective=, verbose=)``               (R:57-61)
``lgb.cv(params, data, nrounds,     ``lgb_cv(params, data, nrounds, nfold,
nfold, objective =, eval =,         objective=, eval=, showsd=,
showsd =, early_stopping_rounds =,  early_stopping_rounds=, verbose=)``   (R:70-78)
verbose =)``
``as.list(cvm)$best_iter``          ``cvm.best_iter`` / ``as_list(cvm)``  (R:79-81)
``predict(fit, X)``                 ``predict(fit, X)``                   (R:63)
================================  =============================================

Semantics kept from the R binding the reference ran against:

* extra named arguments (``objective = "regression"``) are merged into ``params``;
  a parameter row of a grid (all doubles, incl. ``nthread``) is accepted as is (R:108, R:134);
* ``best_score`` is sign-flipped for losses, "high values are good"
  ("LightGBM R.ipynb":443): with ``eval = "rmse"`` it is ``-RMSE``, with the default
  regression metric it is ``-L2``;
* ``compat_sentinel=True`` reproduces the 2017 binding's ``best_iter = best_score = -1`` when
  early stopping never fires within ``nrounds`` (28 rows of the saved ``paramGrid.RData``);
  the default records the arg-best iteration instead.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

from . import callback as cb
from .basic import Booster, Dataset
from .engine import cv as _cv
from .engine import train as _train

__all__ = ["lgb_Dataset", "lgb_train", "lgb_cv", "RCVBooster", "as_list", "predict", "lgb_Dataset_create_valid",
           "lgb_Dataset_construct", "lgb_Dataset_save", "getinfo", "setinfo", "dim", "slice_rows", "lgb_save",
           "lgb_load", "lgb_dump", "lgb_importance", "lgb_interprete", "lgb_model_dt_tree", "lgb_get_eval_result",
           "saveRDS_lgb_Booster", "readRDS_lgb_Booster", "lgb_prepare", "lgb_prepare_rules", "lgb_Dataset_set_categorical",
           "lgb_Dataset_set_reference", "dimnames", "lgb_plot_importance", "lgb_plot_interpretation", "lgb_unloader"]


def lgb_Dataset(data, label=None, weight=None, **kwargs) -> Dataset:
    return Dataset(data, label=label, weight=weight, free_raw_data=kwargs.pop("free_raw_data", False), **kwargs)


def _row_to_params(params) -> Dict[str, Any]:
    """list / dict / 1-row DataFrame / Series -> plain dict (what ``as.list(paramGrid[i, -(1:2)])`` yields)."""
    if params is None:
        return {}
    if hasattr(params, "iloc") and hasattr(params, "columns"):     # DataFrame
        if len(params) != 1:
            raise ValueError("a parameter data frame must have exactly one row")
        params = params.iloc[0]
    if hasattr(params, "to_dict"):
        params = params.to_dict()
    out = {}
    for k, v in dict(params).items():
        if isinstance(v, (np.floating, float)) and float(v).is_integer() and k not in (
                "learning_rate", "feature_fraction", "bagging_fraction", "lambda_l1", "lambda_l2",
                "min_sum_hessian_in_leaf", "min_gain_to_split"):
            v = int(v)
        elif isinstance(v, np.generic):
            v = v.item()
        out[k] = v
    return out


def _verbosity(verbose) -> int:
    return -1 if verbose is None or verbose <= 0 else int(verbose)


def lgb_train(params=None, data: Dataset = None, nrounds: int = 10, valids: Optional[Dict[str, Dataset]] = None,
              obj=None, eval=None, verbose: int = 1, record: bool = True, eval_freq: int = 1, init_model=None,
              early_stopping_rounds: Optional[int] = None, callbacks: Optional[list] = None, **kwargs) -> Booster:
    p = _row_to_params(params)
    p.update(kwargs)
    if isinstance(obj, str):
        p["objective"] = obj
    if isinstance(eval, (str, list, tuple)):
        p["metric"] = eval
    p.setdefault("verbose", _verbosity(verbose))
    nrounds = int(nrounds)
    if nrounds < 1:
        raise ValueError("nrounds must be >= 1 (a grid row that never early-stopped has iteration -1: refit it with "
                         "an explicit number of rounds)")
    valid_sets, valid_names = None, None
    if valids:
        valid_names = list(valids.keys())
        valid_sets = [valids[k] for k in valid_names]
    evals_result: Dict[str, Any] = {}
    bst = _train(p, data, num_boost_round=nrounds, valid_sets=valid_sets, valid_names=valid_names,
                 fobj=obj if callable(obj) else None, feval=eval if callable(eval) else None, init_model=init_model,
                 early_stopping_rounds=early_stopping_rounds, evals_result=evals_result if record else None,
                 verbose_eval=eval_freq if verbose and verbose > 0 else False, callbacks=callbacks)
    bst.record_evals = evals_result
    bst.best_iter = bst.best_iteration if bst.best_iteration > 0 else -1
    return bst


class RCVBooster:
    """Result of :func:`lgb_cv` (``lgb.CVBooster``): ``best_iter``, ``best_score``, ``record_evals``, ``boosters``."""

    def __init__(self):
        self.boosters: List[Booster] = []
        self.best_iter: int = -1
        self.best_score: float = -1.0
        self.record_evals: Dict[str, Any] = {}

    def as_list(self) -> Dict[str, Any]:
        return {"best_iter": self.best_iter, "best_score": self.best_score, "record_evals": self.record_evals,
                "boosters": self.boosters}

    def __getitem__(self, key):
        return self.as_list()[key]


def as_list(x) -> Dict[str, Any]:
    return x.as_list()


def lgb_cv(params=None, data: Dataset = None, nrounds: int = 10, nfold: int = 3, label=None, weight=None, obj=None,
           eval=None, verbose: int = 1, record: bool = True, eval_freq: int = 1, showsd: bool = True,
           stratified: bool = True, folds=None, init_model=None, early_stopping_rounds: Optional[int] = None,
           callbacks: Optional[list] = None, seed: int = 0, compat_sentinel: bool = False,
           lockstep_chunk: Optional[int] = None, **kwargs) -> RCVBooster:
    from .engine import drive_lockstep
    return drive_lockstep(lgb_cv_steps(params, data, nrounds, nfold, label, weight, obj, eval, verbose, record, eval_freq, showsd,
                                       stratified, folds, init_model, early_stopping_rounds, callbacks, seed, compat_sentinel,
                                       lockstep_chunk, **kwargs))


def lgb_cv_steps(params=None, data: Dataset = None, nrounds: int = 10, nfold: int = 3, label=None, weight=None, obj=None,
                 eval=None, verbose: int = 1, record: bool = True, eval_freq: int = 1, showsd: bool = True,
                 stratified: bool = True, folds=None, init_model=None, early_stopping_rounds: Optional[int] = None,
                 callbacks: Optional[list] = None, seed: int = 0, compat_sentinel: bool = False,
                 lockstep_chunk: Optional[int] = None, rollback_overrun: bool = True, **kwargs):
    """:func:`lgb_cv` as a generator of lockstep requests (:func:`lightgbm_amd.engine.cv_steps`)."""
    from .engine import cv_steps
    p = _row_to_params(params)
    p.update(kwargs)
    if isinstance(obj, str):
        p["objective"] = obj
    if isinstance(eval, (str, list, tuple)):
        p["metric"] = eval
    p.setdefault("verbose", _verbosity(verbose))
    if label is not None:
        data.set_label(label)
    if weight is not None:
        data.set_weight(weight)
    res = yield from cv_steps(p, data, num_boost_round=int(nrounds), folds=folds, nfold=int(nfold), stratified=stratified,
                              fobj=obj if callable(obj) else None, feval=eval if callable(eval) else None,
                              early_stopping_rounds=early_stopping_rounds,
                              verbose_eval=eval_freq if verbose and verbose > 0 else None,
                              show_stdv=showsd, seed=seed, callbacks=callbacks, return_cvbooster=True,
                              lockstep_chunk=lockstep_chunk, rollback_overrun=rollback_overrun)
    cvb = res.pop("cvbooster")
    out = RCVBooster()
    out.boosters = cvb.boosters
    # record_evals$valid$<metric>$eval / eval_err, like the R object
    rec: Dict[str, Any] = {"valid": {}}
    first_key = None
    for key, vals in res.items():
        if key.endswith("-mean"):
            m = key[:-5]
            first_key = first_key or m
            rec["valid"].setdefault(m, {})["eval"] = list(vals)
        elif key.endswith("-stdv"):
            rec["valid"].setdefault(key[:-5], {})["eval_err"] = list(vals) if showsd else []
    out.record_evals = rec if record else {}
    if first_key is not None:
        means = np.asarray(res[f"{first_key}-mean"], dtype=np.float64)
        # the engine's own flag for the metric (covers average_precision, auc_mu, custom metrics, ...)
        higher_better = cvb.higher_better.get(first_key, first_key == "auc" or first_key.startswith(("ndcg@", "map@")))
        factor = 1.0 if higher_better else -1.0
        bi = cvb.best_iteration if cvb.best_iteration > 0 else int(np.argmax(factor * means)) + 1
        # the 2017 R callback only recorded a best iteration when the patience criterion fired
        never_fired = (early_stopping_rounds is not None and cvb.iterations_run >= int(nrounds)
                       and (int(nrounds) - bi) < int(early_stopping_rounds))
        if compat_sentinel and never_fired:
            out.best_iter, out.best_score = -1, -1.0
        else:
            out.best_iter = int(bi)
            out.best_score = float(factor * means[bi - 1])
    return out


def predict(booster: Booster, data, num_iteration: Optional[int] = None, rawscore: bool = False,
            predleaf: bool = False, predcontrib: bool = False, **kwargs) -> np.ndarray:
    """``predict(fit, X)`` on a dense matrix (all iterations unless ``num_iteration`` is given)."""
    return booster.predict(data, num_iteration=num_iteration if num_iteration is not None else -1,
                           raw_score=rawscore, pred_leaf=predleaf, pred_contrib=predcontrib, **kwargs)


# ---------------------------------------------------------------------------------------------------------
# The rest of the R binding's everyday surface (not called by the reference scripts, but what a user of
# ``library(lightgbm)`` reaches for next): dataset helpers, model files, importance / interpretation tables.
# ---------------------------------------------------------------------------------------------------------
def lgb_Dataset_create_valid(dataset: Dataset, data, label=None, **kwargs) -> Dataset:
    """``lgb.Dataset.create.valid``: a validation set binned with the training set's boundaries."""
    return dataset.create_valid(data, label=label, **kwargs)


def lgb_Dataset_construct(dataset: Dataset) -> Dataset:
    return dataset.construct()


def lgb_Dataset_save(dataset: Dataset, fname: str) -> Dataset:
    """``lgb.Dataset.save``: binary cache of the binned dataset (reload with ``lgb_Dataset(fname)``)."""
    return dataset.save_binary(fname)


def getinfo(dataset: Dataset, name: str):
    getter = {"label": dataset.get_label, "weight": dataset.get_weight, "init_score": dataset.get_init_score,
              "group": dataset.get_group}
    if name not in getter:
        raise ValueError("getinfo: name must be one of label, weight, init_score, group")
    return getter[name]()


def setinfo(dataset: Dataset, name: str, info) -> Dataset:
    setter = {"label": dataset.set_label, "weight": dataset.set_weight, "init_score": dataset.set_init_score,
              "group": dataset.set_group}
    if name not in setter:
        raise ValueError("setinfo: name must be one of label, weight, init_score, group")
    return setter[name](info)


def dim(dataset: Dataset):
    dataset.construct()
    return (dataset.num_data(), dataset.num_feature())


def slice_rows(dataset: Dataset, idxset) -> Dataset:
    """``slice(dataset, idxset)`` with R's 1-based row numbers."""
    idx = np.asarray(list(idxset), dtype=np.int64) - 1
    return dataset.construct().subset(idx)


def lgb_save(booster: Booster, filename: str, num_iteration: Optional[int] = None) -> Booster:
    return booster.save_model(filename, num_iteration=num_iteration)


def lgb_load(filename: Optional[str] = None, model_str: Optional[str] = None) -> Booster:
    if filename is None and model_str is None:
        raise ValueError("lgb_load: need filename or model_str")
    return Booster(model_file=filename) if filename is not None else Booster(model_str=model_str)


def lgb_dump(booster: Booster, num_iteration: Optional[int] = None) -> str:
    """``lgb.dump``: the model as a JSON string."""
    import json
    return json.dumps(booster.dump_model(num_iteration=num_iteration))


_RDS_MAGIC = "lightgbm_amd booster v1"


def saveRDS_lgb_Booster(booster: Booster, file: str) -> None:
    """``saveRDS.lgb.Booster``: the booster as one self-contained file - a JSON header line (best iteration,
    best scores, parameters) followed by the text model.  Nothing in the file is executable."""
    import json
    header = {"format": _RDS_MAGIC, "best_iteration": int(booster.best_iteration),
              "best_score": getattr(booster, "best_score", {}), "params": dict(booster.params)}
    with open(file, "w") as f:
        f.write(json.dumps(header, default=str) + "\n")
        f.write(booster.model_to_string(num_iteration=-1))


def readRDS_lgb_Booster(file: str) -> Booster:
    """Reads a file written by :func:`saveRDS_lgb_Booster` (JSON header + text model; no unpickling)."""
    import json
    with open(file, "r") as f:
        first = f.readline()
        try:
            header = json.loads(first)
        except ValueError:
            header = None
        if not isinstance(header, dict) or header.get("format") != _RDS_MAGIC:
            raise ValueError(f"{file}: not a booster file written by saveRDS_lgb_Booster")
        model = f.read()
    params = header.get("params") if isinstance(header.get("params"), dict) else None
    b = Booster(params=params, model_str=model)
    b.best_iteration = int(header.get("best_iteration", 0))
    b.best_score = header.get("best_score", {}) if isinstance(header.get("best_score"), dict) else {}
    return b


def lgb_importance(booster: Booster, percentage: bool = True):
    """``lgb.importance``: data frame ``Feature, Gain, Cover, Frequency`` sorted by gain (``Cover`` = rows that
    pass through the feature's splits)."""
    import pandas as pd
    df = booster.trees_to_dataframe()
    splits = df[df["split_feature"].notna()]
    agg = splits.groupby("split_feature").agg(Gain=("split_gain", "sum"), Cover=("count", "sum"),
                                              Frequency=("split_gain", "size")).reset_index()
    agg = agg.rename(columns={"split_feature": "Feature"})
    for c in ("Gain", "Cover", "Frequency"):
        agg[c] = agg[c].astype(float)
        if percentage and agg[c].sum() > 0:
            agg[c] = agg[c] / agg[c].sum()
    return agg.sort_values("Gain", ascending=False).reset_index(drop=True)


def lgb_interprete(booster: Booster, data, idxset, num_iteration: Optional[int] = None):
    """``lgb.interprete``: per requested row (1-based) a data frame ``Feature, Contribution`` sorted by absolute
    contribution; the contributions are SHAP values in raw-score units."""
    import pandas as pd
    X = np.asarray(data)
    idx = np.asarray(list(idxset), dtype=np.int64) - 1
    phi = booster.predict(X[idx], pred_contrib=True, num_iteration=num_iteration if num_iteration else -1)
    names = booster.feature_name()
    out = []
    for r in range(len(idx)):
        df = pd.DataFrame({"Feature": names, "Contribution": phi[r, : len(names)]})
        out.append(df.reindex(df["Contribution"].abs().sort_values(ascending=False).index).reset_index(drop=True))
    return out


def lgb_model_dt_tree(booster: Booster, num_iteration: Optional[int] = None):
    """``lgb.model.dt.tree``: one row per node, R's column names."""
    df = booster.trees_to_dataframe()
    if num_iteration is not None and num_iteration > 0:
        df = df[df["tree_index"] < num_iteration]
    return df.rename(columns={"node_depth": "depth", "parent_index": "node_parent"})


def lgb_get_eval_result(booster, data_name: str, eval_name: str, iters=None, is_err: bool = False):
    """``lgb.get.eval.result``: recorded metric values of a ``lgb_train`` / ``lgb_cv`` result."""
    rec = getattr(booster, "record_evals", None) or {}
    if data_name not in rec:
        raise KeyError(f"data_name {data_name!r} not in record_evals ({sorted(rec)})")
    if eval_name not in rec[data_name]:
        raise KeyError(f"eval_name {eval_name!r} not in record_evals[{data_name!r}] ({sorted(rec[data_name])})")
    series = rec[data_name][eval_name]
    if isinstance(series, dict):
        series = series["eval_err" if is_err else "eval"]
    vals = np.asarray(series, dtype=np.float64)
    if iters is not None:
        vals = vals[np.asarray(list(iters), dtype=np.int64) - 1]
    return vals


# ---------------------------------------------------------------------------------------------------------
# Data preparation (``lgb.prepare`` / ``lgb.prepare_rules``): the reference turns its factor columns into
# integers by hand before building the matrix (r/gridsearchCV.R:11-13); these helpers do the same for a
# pandas data frame.
# ---------------------------------------------------------------------------------------------------------
def lgb_prepare_rules(data, rules: Optional[Dict[str, Dict[Any, int]]] = None):
    """Converts the non-numeric columns (category / object / bool) of a data frame to integer codes 1, 2, ...
    Returns ``{"data": converted frame, "rules": {column: {level: code}}}``; passing the rules of an earlier
    call applies the same codes to new data (levels without a code become NaN)."""
    import pandas as pd
    out = data.copy()
    new_rules: Dict[str, Dict[Any, int]] = {}
    for col in out.columns:
        s = out[col]
        if rules is not None:
            if col in rules:
                out[col] = s.map(rules[col]).astype(np.float64)
                new_rules[col] = dict(rules[col])
            continue
        if isinstance(s.dtype, pd.CategoricalDtype):
            levels = list(s.cat.categories)
        elif s.dtype == object or s.dtype == bool or pd.api.types.is_string_dtype(s.dtype):
            levels = sorted(s.dropna().unique().tolist())
        else:
            continue
        mapping = {lev: i + 1 for i, lev in enumerate(levels)}
        out[col] = s.map(mapping).astype(np.float64)
        new_rules[col] = mapping
    return {"data": out, "rules": new_rules}


def lgb_prepare(data):
    """``lgb.prepare``: the converted frame only (see ``lgb_prepare_rules``)."""
    return lgb_prepare_rules(data)["data"]


def lgb_Dataset_set_categorical(dataset: Dataset, categorical_feature) -> Dataset:
    """``lgb.Dataset.set.categorical``: columns (indices or names) holding category codes; before construction only."""
    if dataset._handle is not None:
        raise ValueError("lgb_Dataset_set_categorical: the dataset is already constructed; create a new one")
    dataset.categorical_feature = list(categorical_feature)
    return dataset


def lgb_Dataset_set_reference(dataset: Dataset, reference: Dataset) -> Dataset:
    """``lgb.Dataset.set.reference``: bin ``dataset`` with ``reference``'s boundaries; before construction only."""
    if dataset._handle is not None:
        raise ValueError("lgb_Dataset_set_reference: the dataset is already constructed; create a new one")
    dataset.reference = reference
    return dataset


def dimnames(dataset: Dataset):
    """``dimnames(dataset)``: ``(None, column names)``."""
    return (None, dataset.construct().get_feature_name())


def lgb_plot_importance(tree_imp, top_n: int = 10, measure: str = "Gain", ax=None, **kwargs):
    """``lgb.plot.importance``: bars of one measure (``Gain`` / ``Cover`` / ``Frequency``) of an ``lgb_importance`` table."""
    import matplotlib.pyplot as plt
    if measure not in tree_imp.columns:
        raise ValueError(f"measure must be one of {[c for c in tree_imp.columns if c != 'Feature']}")
    top = tree_imp.sort_values(measure, ascending=False).head(int(top_n)).iloc[::-1]
    if ax is None:
        _, ax = plt.subplots(1, 1)
    ax.barh(top["Feature"].astype(str), top[measure], **kwargs)
    ax.set_xlabel(measure)
    ax.set_title("Feature Importance")
    return ax


def lgb_plot_interpretation(tree_interpretation, top_n: int = 10, ax=None, **kwargs):
    """``lgb.plot.interpretation``: the largest per-feature contributions of one row of ``lgb_interprete``."""
    import matplotlib.pyplot as plt
    df = tree_interpretation
    col = [c for c in df.columns if c != "Feature"][0]
    top = df.reindex(df[col].abs().sort_values(ascending=False).index).head(int(top_n)).iloc[::-1]
    if ax is None:
        _, ax = plt.subplots(1, 1)
    ax.barh(top["Feature"].astype(str), top[col], color=["#d62728" if v < 0 else "#1f77b4" for v in top[col]], **kwargs)
    ax.set_xlabel(col)
    ax.set_title("Feature Contribution")
    return ax


def lgb_unloader(restore: bool = True, wipe: bool = False, envir: Optional[Dict[str, Any]] = None) -> None:
    """``lgb.unloader``: drops the boosters / datasets held in ``envir`` (a dict of names, e.g. ``globals()``)
    when ``wipe`` is set, and collects garbage so that their native handles are released."""
    import gc
    if wipe and envir is not None:
        for name in [k for k, v in envir.items() if isinstance(v, (Booster, Dataset, RCVBooster))]:
            del envir[name]
    gc.collect()
