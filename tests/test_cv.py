"""k-fold CV: folds, lock-step boosters, early stopping on the fold mean, R-style result object."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.engine import make_folds
from lightgbm_amd.rstyle import as_list, lgb_cv, lgb_Dataset, lgb_train, predict
from lightgbm_amd.utils.datasets import make_diamonds_like, make_regression

CPU = {"device_type": "cpu", "verbose": -1}


def test_folds_are_disjoint_covering_and_balanced():
    y = np.random.RandomState(0).randn(1003)
    for stratified in (True, False):
        folds = make_folds(1003, 5, y, stratified=stratified, seed=3)
        test_all = np.concatenate([te for _, te in folds])
        assert sorted(test_all.tolist()) == list(range(1003))
        for tr, te in folds:
            assert len(np.intersect1d(tr, te)) == 0 and len(tr) + len(te) == 1003
            assert 195 <= len(te) <= 206
    # stratified on a numeric label: every fold sees every label quantile group
    folds = make_folds(1003, 5, y, stratified=True, seed=1)
    cuts = np.quantile(y, [0.2, 0.4, 0.6, 0.8])
    for _, te in folds:
        grp = np.searchsorted(cuts, y[te])
        assert np.bincount(grp, minlength=5).min() >= 30
    # classification: per-class stratification
    yc = (np.random.RandomState(1).rand(1000) < 0.1).astype(float)
    for _, te in make_folds(1000, 5, yc, stratified=True, seed=0, classification=True):
        assert abs(yc[te].sum() - yc.sum() / 5) <= 1
    # seeds change the folds, equal seeds reproduce them
    a = make_folds(100, 5, None, stratified=False, seed=1)
    b = make_folds(100, 5, None, stratified=False, seed=1)
    c = make_folds(100, 5, None, stratified=False, seed=2)
    assert all(np.array_equal(x[1], z[1]) for x, z in zip(a, b))
    assert any(not np.array_equal(x[1], z[1]) for x, z in zip(a, c))


def test_cv_returns_mean_and_stdv_and_stops_early():
    X, y = make_regression(4000, 8, seed=2, noise=1.0)
    res = lgb.cv({**CPU, "learning_rate": 0.3, "metric": "rmse"}, lgb.Dataset(X, label=y), num_boost_round=400,
                 nfold=5, early_stopping_rounds=5, seed=7, return_cvbooster=True)
    mean, std = res["rmse-mean"], res["rmse-stdv"]
    cvb = res["cvbooster"]
    assert len(mean) == len(std) == cvb.best_iteration < 400
    assert cvb.best_iteration == int(np.argmin(mean)) + 1
    assert len(cvb.boosters) == 5 and all(s >= 0 for s in std)
    # every fold booster trained best_iteration + patience rounds in lock step
    assert all(b.num_trees() == cvb.best_iteration + 5 for b in cvb.boosters)


def test_cv_fold_metric_equals_manual_holdout():
    X, y = make_regression(2000, 5, seed=4)
    folds = make_folds(2000, 4, y, stratified=False, seed=0)
    res = lgb.cv({**CPU, "num_leaves": 15}, lgb.Dataset(X, label=y), num_boost_round=10, folds=folds)
    manual = []
    for tr, te in folds:
        d = lgb.Dataset(X, label=y)
        b = lgb.train({**CPU, "num_leaves": 15}, d.subset(tr), 10)
        manual.append(np.mean((b.predict(X[te]) - y[te]) ** 2))
    assert abs(res["l2-mean"][-1] - np.mean(manual)) < 1e-6 * np.mean(manual) + 1e-9


def test_r_style_cv_sign_flip_and_metric_choice():
    # NBR:443 "high values are good": best_score = -RMSE with eval="rmse" (R:75,81), -L2 by default (R:108-117)
    X, y = make_diamonds_like(8000, seed=1)
    d = lgb_Dataset(X, label=y, params=CPU)
    a = lgb_cv({"learning_rate": 0.1}, d, nrounds=300, nfold=5, objective="regression", eval="rmse", showsd=False,
               early_stopping_rounds=5, verbose=0, seed=1)
    b = lgb_cv({"learning_rate": 0.1}, d, nrounds=300, nfold=5, objective="regression", showsd=False,
               early_stopping_rounds=5, verbose=0, seed=1)
    assert a.best_iter > 10 and a.best_score < 0 and b.best_score < 0
    assert abs((-a.best_score) ** 2 - (-b.best_score)) < 0.05 * (-b.best_score)
    info = as_list(a)
    assert info["best_iter"] == a.best_iter and "rmse" in info["record_evals"]["valid"]
    hist = info["record_evals"]["valid"]["rmse"]["eval"]
    assert len(hist) == a.best_iter and abs(hist[-1] + a.best_score) < 1e-12
    assert info["record_evals"]["valid"]["rmse"]["eval_err"] == []          # showsd = FALSE


def test_r_style_sentinel_compat():
    # rows of the saved grid that never early-stopped within nrounds hold iteration = score = -1
    X, y = make_diamonds_like(3000, seed=2)
    d = lgb_Dataset(X, label=y, params=CPU)
    kw = dict(nrounds=30, nfold=3, objective="regression", showsd=False, early_stopping_rounds=5, verbose=0)
    old = lgb_cv({"learning_rate": 0.01}, d, compat_sentinel=True, **kw)
    new = lgb_cv({"learning_rate": 0.01}, d, **kw)
    assert (old.best_iter, old.best_score) == (-1, -1.0)
    assert new.best_iter == 30 and new.best_score < 0
    with pytest.raises(ValueError, match="nrounds"):
        lgb_train({"learning_rate": 0.1}, d, nrounds=old.best_iter, objective="regression", verbose=0)


def test_r_style_train_accepts_dataframe_row():
    import pandas as pd
    X, y = make_diamonds_like(3000, seed=3)
    d = lgb_Dataset(X, label=y, params=CPU)
    row = pd.DataFrame([{"learning_rate": 0.1, "num_leaves": 31.0, "min_data_in_leaf": 20.0, "feature_fraction": 1.0,
                         "bagging_fraction": 0.8, "bagging_freq": 4.0, "nthread": 4.0}])
    fit = lgb_train(row, d, nrounds=20, objective="regression", verbose=0)
    assert fit.num_trees() == 20
    assert np.sqrt(np.mean((predict(fit, X) - y) ** 2)) < 0.3


def test_register_logger_receives_the_messages():
    import lightgbm_amd as lgb

    class Sink:
        def __init__(self):
            self.info_lines, self.warn_lines = [], []

        def info(self, msg):
            self.info_lines.append(msg)

        def warning(self, msg):
            self.warn_lines.append(msg)
    sink = Sink()
    lgb.register_logger(sink)
    try:
        rng = np.random.RandomState(0)
        X = rng.randn(600, 3)
        y = X[:, 0] + 0.1 * rng.randn(600)
        d = lgb.Dataset(X, label=y, params={"device_type": "cpu", "verbose": -1})
        lgb.train({"device_type": "cpu", "verbose": 1, "no_such_parameter": 1}, d, 30, valid_sets=[d], early_stopping_rounds=3,
                  verbose_eval=10)
    finally:
        lgb.register_logger(None)
    assert any("no_such_parameter" in m for m in sink.warn_lines)
    assert any(m.startswith("[10]") for m in sink.info_lines)
    import pytest
    with pytest.raises(TypeError):
        lgb.register_logger(object())


def test_fold_statistics_in_plain_python_equal_numpys_bit_for_bit():
    """cv aggregates its handful of fold values without numpy's per-call overhead (engine._mean_std); the table of a
    grid search must not depend on that shortcut."""
    from lightgbm_amd.engine import _mean_std
    rng = np.random.default_rng(5)
    for _ in range(20000):
        n = int(rng.integers(1, 12))
        v = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6)).tolist()
        m, s = _mean_std(v)
        assert m == float(np.mean(v)) and s == float(np.std(v))


def test_cached_fold_draws_follow_the_labels_and_the_arguments():
    """cv keeps the folds it drew on the dataset (a grid search draws the same folds for every row); new labels,
    another seed or another fold count must draw anew, and a repeated call must give the same numbers."""
    X, y = make_regression(600, 5, seed=3)
    params = {"objective": "regression", "num_leaves": 7, "min_data_in_leaf": 5, "verbose": -1, "device_type": "cpu"}
    d = lgb.Dataset(X, label=y, params=params, free_raw_data=False)
    a = lgb.cv(params, d, num_boost_round=5, nfold=3, stratified=False, seed=1)
    assert len(d._fold_draws) == 1
    b = lgb.cv(params, d, num_boost_round=5, nfold=3, stratified=False, seed=1)
    assert a == b and len(d._fold_draws) == 1
    c = lgb.cv(params, d, num_boost_round=5, nfold=3, stratified=False, seed=2)
    assert len(d._fold_draws) == 2 and c != a
    lgb.cv(params, d, num_boost_round=5, nfold=4, stratified=False, seed=1)
    assert len(d._fold_draws) == 3
    # the same arguments on a fresh dataset: what the cache returns is what a first draw gives
    d2 = lgb.Dataset(X, label=y, params=params, free_raw_data=False)
    assert lgb.cv(params, d2, num_boost_round=5, nfold=3, stratified=False, seed=1) == a
    # new labels under the same key: drawn (and computed) anew
    d.set_label(y[::-1].copy())
    e = lgb.cv(params, d, num_boost_round=5, nfold=3, stratified=False, seed=1)
    d3 = lgb.Dataset(X, label=y[::-1].copy(), params=params, free_raw_data=False)
    assert e == lgb.cv(params, d3, num_boost_round=5, nfold=3, stratified=False, seed=1)
