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

__all__ = ["lgb_Dataset", "lgb_train", "lgb_cv", "RCVBooster", "as_list", "predict"]


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
           callbacks: Optional[list] = None, seed: int = 0, compat_sentinel: bool = False, **kwargs) -> RCVBooster:
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
    res = _cv(p, data, num_boost_round=int(nrounds), folds=folds, nfold=int(nfold), stratified=stratified,
              fobj=obj if callable(obj) else None, feval=eval if callable(eval) else None,
              early_stopping_rounds=early_stopping_rounds, verbose_eval=eval_freq if verbose and verbose > 0 else None,
              show_stdv=showsd, seed=seed, callbacks=callbacks, return_cvbooster=True)
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
        higher_better = first_key in ("auc",)
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


def predict(booster: Booster, data, num_iteration: Optional[int] = None, rawscore: bool = False, **kwargs) -> np.ndarray:
    """``predict(fit, X)`` on a dense matrix (all iterations unless ``num_iteration`` is given)."""
    return booster.predict(data, num_iteration=num_iteration if num_iteration is not None else -1,
                           raw_score=rawscore, **kwargs)
