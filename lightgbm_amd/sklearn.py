"""scikit-learn style estimators (``LGBMRegressor``, ``LGBMClassifier``, ``LGBMRanker``, ``LGBMRandomForestRegressor``).

bagging_boosting.ipynb:204 drives its forest through the sklearn estimator interface
(``RandomForestRegressor(n_estimators, max_leaf_nodes, ...).fit(x, y).predict(x)``); these wrappers
give the same calling style on top of :func:`lightgbm_amd.train`.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from .basic import Booster, Dataset
from .engine import train

try:   # sklearn is optional
    from sklearn.base import BaseEstimator, ClassifierMixin, RegressorMixin
except Exception:   # pragma: no cover
    class BaseEstimator:   # type: ignore
        pass

    class RegressorMixin:   # type: ignore
        pass

    class ClassifierMixin:   # type: ignore
        pass

__all__ = ["LGBMModel", "LGBMRegressor", "LGBMClassifier", "LGBMRanker", "LGBMRandomForestRegressor"]


class LGBMModel(BaseEstimator):
    def __init__(self, boosting_type: str = "gbdt", num_leaves: int = 31, max_depth: int = -1,
                 learning_rate: float = 0.1, n_estimators: int = 100, subsample_for_bin: int = 200000,
                 objective: Optional[str] = None, min_split_gain: float = 0.0, min_child_weight: float = 1e-3,
                 min_child_samples: int = 20, subsample: float = 1.0, subsample_freq: int = 0,
                 colsample_bytree: float = 1.0, reg_alpha: float = 0.0, reg_lambda: float = 0.0,
                 random_state: Optional[int] = None, n_jobs: int = 0, max_bin: int = 255, device_type: Optional[str] = None,
                 verbose: int = -1, **kwargs):
        self.boosting_type = boosting_type
        self.num_leaves = num_leaves
        self.max_depth = max_depth
        self.learning_rate = learning_rate
        self.n_estimators = n_estimators
        self.subsample_for_bin = subsample_for_bin
        self.objective = objective
        self.min_split_gain = min_split_gain
        self.min_child_weight = min_child_weight
        self.min_child_samples = min_child_samples
        self.subsample = subsample
        self.subsample_freq = subsample_freq
        self.colsample_bytree = colsample_bytree
        self.reg_alpha = reg_alpha
        self.reg_lambda = reg_lambda
        self.random_state = random_state
        self.n_jobs = n_jobs
        self.max_bin = max_bin
        self.device_type = device_type
        self.verbose = verbose
        self._other_params: Dict[str, Any] = dict(kwargs)
        self._Booster: Optional[Booster] = None
        self.best_iteration_ = 0
        self.evals_result_: Dict[str, Any] = {}

    # extra keyword parameters (any training parameter) are part of the estimator's parameter set, so that
    # clone() / GridSearchCV / set_params see them like the named ones
    def get_params(self, deep: bool = True) -> Dict[str, Any]:
        params = super().get_params(deep=deep)
        params.update(self._other_params)
        return params

    def set_params(self, **params) -> "LGBMModel":
        for key, value in params.items():
            if hasattr(self, key) and key in super().get_params(deep=False):
                setattr(self, key, value)
            else:
                self._other_params[key] = value
        return self

    def _default_objective(self) -> str:
        return "regression"

    def _params(self) -> Dict[str, Any]:
        p = {
            "boosting": self.boosting_type, "num_leaves": self.num_leaves, "max_depth": self.max_depth,
            "learning_rate": self.learning_rate, "bin_construct_sample_cnt": self.subsample_for_bin,
            "objective": self.objective or self._default_objective(), "min_gain_to_split": self.min_split_gain,
            "min_sum_hessian_in_leaf": self.min_child_weight, "min_data_in_leaf": self.min_child_samples,
            "bagging_fraction": self.subsample, "bagging_freq": self.subsample_freq,
            "feature_fraction": self.colsample_bytree, "lambda_l1": self.reg_alpha, "lambda_l2": self.reg_lambda,
            "num_threads": self.n_jobs, "max_bin": self.max_bin, "verbose": self.verbose,
        }
        if self.random_state is not None:
            p["seed"] = int(self.random_state)
            p["bagging_seed"] = int(self.random_state)
            p["feature_fraction_seed"] = int(self.random_state)
        if self.device_type is not None:
            p["device_type"] = self.device_type
        p.update(self._other_params)
        p.pop("class_weight", None)      # an estimator-level setting (turned into row weights), not a training parameter
        return p

    def fit(self, X, y, sample_weight=None, init_score=None, eval_set=None, eval_names=None, eval_metric=None,
            early_stopping_rounds=None, callbacks=None, init_model=None, group=None, eval_group=None,
            eval_sample_weight=None, eval_init_score=None, feature_name="auto", categorical_feature="auto"):
        params = self._params()
        feval = None
        if callable(eval_metric):
            # sklearn-style metric f(y_true, y_pred) -> (name, value, is_higher_better)
            user = eval_metric
            feval = lambda preds, data: user(data.get_label(), preds)   # noqa: E731
        elif eval_metric is not None:
            params["metric"] = eval_metric
        dtrain = Dataset(X, label=self._encode_label(y), weight=sample_weight, init_score=init_score, params=params,
                         free_raw_data=init_model is None, group=group, feature_name=feature_name,
                         categorical_feature=categorical_feature)
        valid_sets, valid_names = [], []
        pick = lambda seq, i: None if seq is None else (seq.get(i) if isinstance(seq, dict) else seq[i])   # noqa: E731
        for i, (vx, vy) in enumerate(eval_set or []):
            if vx is X and vy is y:
                valid_sets.append(dtrain)
            else:
                valid_sets.append(dtrain.create_valid(vx, label=self._encode_label(vy), weight=pick(eval_sample_weight, i),
                                                      init_score=pick(eval_init_score, i), group=pick(eval_group, i)))
            valid_names.append(eval_names[i] if eval_names else f"valid_{i}")
        self.evals_result_ = {}
        self._Booster = train(params, dtrain, num_boost_round=self.n_estimators, valid_sets=valid_sets or None,
                              valid_names=valid_names or None, early_stopping_rounds=early_stopping_rounds,
                              evals_result=self.evals_result_ if valid_sets else None, callbacks=callbacks,
                              init_model=init_model, feval=feval)
        self.best_iteration_ = self._Booster.best_iteration
        self.best_score_ = getattr(self._Booster, "best_score", {})
        self.n_features_in_ = self.n_features_ = self._Booster.num_feature()
        self.feature_name_ = self._Booster.feature_name()
        self.objective_ = params["objective"]
        self.n_iter_ = self._Booster.current_iteration()
        self.fitted_ = True
        return self

    def _encode_label(self, y):
        return y

    @property
    def booster_(self) -> Booster:
        if self._Booster is None:
            raise RuntimeError("Estimator is not fitted")
        return self._Booster

    @property
    def feature_importances_(self) -> np.ndarray:
        return self.booster_.feature_importance()

    def predict(self, X, raw_score: bool = False, num_iteration: Optional[int] = None, pred_leaf: bool = False, **kw):
        return self.booster_.predict(X, raw_score=raw_score, num_iteration=num_iteration, pred_leaf=pred_leaf, **kw)

    def staged_predict(self, X, step: int = 1, **kw):
        """Predictions of the first ``step, 2 * step, ...`` trees - the curves "bagging_boosting.ipynb":118-137
        draws for a growing forest and a growing boosted ensemble (``predict(ntree_limit=k)`` there)."""
        b = self.booster_
        per_iter = max(1, b.num_model_per_iteration())
        for k in range(step, b.num_trees() // per_iter + 1, step):
            yield b.predict(X, num_iteration=k, **kw)


class LGBMRegressor(RegressorMixin, LGBMModel):
    pass


class LGBMRanker(LGBMModel):
    """Learning to rank (``lambdarank`` unless another objective is given): ``fit`` needs ``group``, the sizes
    of the queries in row order; every ``eval_set`` entry needs its ``eval_group``.  ``eval_at`` lists the
    cut-offs of the ndcg / map metrics.  ``predict`` returns relevance scores (rank within a query by them)."""

    def _default_objective(self) -> str:
        return "lambdarank"

    def fit(self, X, y, group=None, eval_set=None, eval_group=None, eval_at=(1, 2, 3, 4, 5), **kwargs):
        if group is None:
            raise ValueError("LGBMRanker.fit needs group (the query sizes)")
        if eval_set is not None and (eval_group is None or len(eval_group) != len(eval_set)):
            raise ValueError("LGBMRanker.fit needs one eval_group entry per eval_set entry")
        self._other_params["eval_at"] = [int(k) for k in eval_at]
        return super().fit(X, y, group=group, eval_set=eval_set, eval_group=eval_group, **kwargs)


class LGBMRandomForestRegressor(RegressorMixin, LGBMModel):
    """Bagged forest: ``boosting='rf'`` with bootstrap-sized bags (0.632) unless told otherwise."""

    def __init__(self, n_estimators: int = 100, num_leaves: int = 31, subsample: float = 0.632, subsample_freq: int = 1,
                 colsample_bytree: float = 1.0, **kwargs):
        kwargs.pop("boosting_type", None)
        super().__init__(boosting_type="rf", n_estimators=n_estimators, num_leaves=num_leaves, subsample=subsample,
                         subsample_freq=subsample_freq, colsample_bytree=colsample_bytree, **kwargs)


class LGBMClassifier(ClassifierMixin, LGBMModel):
    """Classifier: ``binary`` for two classes, ``multiclass`` (softmax, one tree per class and iteration)
    for more.  Labels of any type are mapped to 0 .. n_classes-1 in sorted order (``classes_``)."""

    def _default_objective(self) -> str:
        return "binary" if getattr(self, "n_classes_", 2) <= 2 else "multiclass"

    def _params(self) -> Dict[str, Any]:
        p = super()._params()
        if getattr(self, "n_classes_", 2) > 2:
            if p["objective"] in ("binary",):
                p["objective"] = "multiclass"
            p["num_class"] = int(self.n_classes_)
        return p

    def _encode_label(self, y):
        y = np.asarray(y)
        if self.n_classes_ <= 2:
            return (y == self.classes_[1]).astype(np.float64)
        return np.searchsorted(self.classes_, y).astype(np.float64)

    def fit(self, X, y, sample_weight=None, **kwargs):
        self.classes_ = np.unique(np.asarray(y))
        self.n_classes_ = len(self.classes_)
        if self.n_classes_ < 2:
            raise ValueError("LGBMClassifier needs at least two classes")
        cw = self._other_params.get("class_weight")
        if cw is not None:
            # "balanced": n / (n_classes * count(class)); a dict: weight per class label (missing labels: 1)
            ya = np.asarray(y)
            if isinstance(cw, str):
                if cw != "balanced":
                    raise ValueError("class_weight must be 'balanced' or a dict")
                counts = np.array([(ya == c).sum() for c in self.classes_], dtype=np.float64)
                per_class = len(ya) / (self.n_classes_ * counts)
            else:
                per_class = np.array([float(cw.get(c.item() if hasattr(c, "item") else c, 1.0)) for c in self.classes_])
            w = per_class[np.searchsorted(self.classes_, ya)]
            sample_weight = w if sample_weight is None else w * np.asarray(sample_weight, dtype=np.float64)
        return super().fit(X, y, sample_weight=sample_weight, **kwargs)

    def predict_proba(self, X, **kw) -> np.ndarray:
        p = self.booster_.predict(X, **kw)
        return p if p.ndim == 2 else np.column_stack([1.0 - p, p])

    def predict(self, X, raw_score: bool = False, **kw):
        if raw_score or kw.get("pred_leaf") or kw.get("pred_contrib"):
            return self.booster_.predict(X, raw_score=raw_score, **kw)
        p = self.booster_.predict(X, **kw)
        if p.ndim == 2:
            return self.classes_[p.argmax(axis=1)]
        return self.classes_[(p > 0.5).astype(int)]
