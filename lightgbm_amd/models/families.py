"""The two model families of the reference as small estimator classes.

* :class:`GBDT` - leaf-wise gradient boosting ("LightGBM R.ipynb":14-22, r/gridsearchCV.R:55-61).
* :class:`RandomForest` - bagged trees, every tree fitted to the same gradients and the
  outputs averaged ("bagging_boosting.ipynb":118-137 builds the same thing from sklearn parts:
  bootstrap samples of ~63 % distinct rows, a feature subset per tree, the mean of the trees).

Both wrap :func:`lightgbm_amd.train`; ``staged_predict`` yields the prediction after
1, 2, ... trees - the curve the notebook plots for bagging against boosting.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, Optional

import numpy as np

from ..basic import Booster, Dataset
from ..engine import train


class _Family:
    boosting = "gbdt"
    defaults: Dict[str, Any] = {}

    def __init__(self, objective: str = "regression", n_estimators: int = 100, **params):
        self.params: Dict[str, Any] = {**self.defaults, **params, "objective": objective, "boosting": self.boosting}
        self.n_estimators = int(n_estimators)
        self.booster_: Optional[Booster] = None

    def fit(self, X, y, sample_weight=None, eval_set=None, early_stopping_rounds: Optional[int] = None, **kw):
        p = dict(self.params)
        p.setdefault("verbose", -1)
        dtrain = Dataset(X, label=y, weight=sample_weight, params=p)
        valid = [dtrain.create_valid(vx, label=vy) for vx, vy in (eval_set or [])]
        self.evals_result_: Dict[str, Any] = {}
        self.booster_ = train(p, dtrain, num_boost_round=self.n_estimators, valid_sets=valid or None,
                              early_stopping_rounds=early_stopping_rounds,
                              evals_result=self.evals_result_ if valid else None, **kw)
        return self

    def _need(self) -> Booster:
        if self.booster_ is None:
            raise RuntimeError("call fit() first")
        return self.booster_

    def predict(self, X, num_iteration: Optional[int] = None, raw_score: bool = False) -> np.ndarray:
        return self._need().predict(X, num_iteration=num_iteration, raw_score=raw_score)

    def staged_predict(self, X, step: int = 1) -> Iterator[np.ndarray]:
        """Predictions of the first ``step, 2*step, ...`` trees."""
        b = self._need()
        for k in range(step, b.num_trees() + 1, step):
            yield b.predict(X, num_iteration=k)

    @property
    def feature_importances_(self) -> np.ndarray:
        return self._need().feature_importance()

    def save_model(self, path: str):
        self._need().save_model(path)


class GBDT(_Family):
    """Gradient-boosted trees: each tree fits the gradients left by the ones before it, scaled by ``learning_rate``."""
    boosting = "gbdt"
    defaults = {"learning_rate": 0.1, "num_leaves": 31}


class RandomForest(_Family):
    """Bagging: independent trees on row / feature sub-samples, averaged (no shrinkage)."""
    boosting = "rf"
    defaults = {"bagging_fraction": 0.632, "bagging_freq": 1, "feature_fraction": 0.8, "num_leaves": 127,
                "min_data_in_leaf": 5}
