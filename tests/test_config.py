"""Parameter registry: aliases, defaults, validation, merging."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.basic import get_param, params_to_string, pop_param, resolve_alias


def test_aliases_used_by_the_reference():
    # r/gridsearchCV.R passes nthread, nrounds (as arg), eval; xgboost-style names appear in the notebook
    assert resolve_alias("nthread") == "num_threads"
    assert resolve_alias("nrounds") == "num_iterations"
    assert resolve_alias("eta") == "learning_rate"
    assert resolve_alias("subsample") == "bagging_fraction"
    assert resolve_alias("colsample_bytree") == "feature_fraction"
    assert resolve_alias("max_leaf_nodes") == "num_leaves"
    assert resolve_alias("random_state") == "seed"
    assert resolve_alias("early_stopping_rounds") == "early_stopping_round"
    assert resolve_alias("no_such_param") is None


def test_params_to_string_and_pop():
    s = params_to_string({"learning_rate": 0.05, "metric": ["l2", "rmse"], "use_missing": False, "skip": None})
    assert "learning_rate=0.05" in s and "metric=l2,rmse" in s and "use_missing=false" in s and "skip" not in s
    p = {"eta": 0.3, "learning_rate": 0.1, "nrounds": 7}
    assert pop_param(p, "learning_rate") == 0.1 and "eta" not in p and "learning_rate" not in p
    assert get_param({"n_estimators": 12}, "num_iterations") == 12


def test_unknown_parameter_warns_not_fails():
    X = np.random.RandomState(0).randn(200, 3)
    y = X[:, 0]
    with pytest.warns(UserWarning, match="Unknown parameter"):
        lgb.train({"verbose": -1, "definitely_unknown": 3, "min_data_in_leaf": 5}, lgb.Dataset(X, label=y), 2)


def test_validation_errors():
    X = np.random.RandomState(0).randn(200, 3)
    y = X[:, 0]
    with pytest.raises(lgb.LightGBMError, match="rf"):
        lgb.train({"boosting": "rf", "verbose": -1}, lgb.Dataset(X, label=y), 2)
    with pytest.raises(lgb.LightGBMError, match="out of range"):
        lgb.train({"bagging_fraction": 1.5, "verbose": -1}, lgb.Dataset(X, label=y), 2)
    with pytest.raises(lgb.LightGBMError, match="objective"):
        lgb.train({"objective": "nonsense", "verbose": -1}, lgb.Dataset(X, label=y), 2)
    with pytest.raises(lgb.LightGBMError, match="integer"):
        lgb.train({"num_leaves": 7.5, "verbose": -1}, lgb.Dataset(X, label=y), 2)


def test_grid_row_of_doubles_is_accepted():
    # as.list(paramGrid[i, -(1:2)]) yields doubles for every column (num_leaves = 31.0, nthread = 4.0)
    X = np.random.RandomState(0).randn(300, 3)
    y = X[:, 0]
    b = lgb.train({"num_leaves": 7.0, "min_data_in_leaf": 5.0, "bagging_freq": 4.0, "nthread": 4.0, "verbose": -1},
                  lgb.Dataset(X, label=y), 3)
    assert b.num_trees() == 3


def test_parameters_block_in_model():
    X = np.random.RandomState(0).randn(300, 3)
    y = X[:, 0]
    b = lgb.train({"learning_rate": 0.07, "num_leaves": 5, "verbose": -1, "min_data_in_leaf": 5}, lgb.Dataset(X, label=y), 2)
    s = b.model_to_string()
    assert "[learning_rate: 0.07" in s and "[num_leaves: 5]" in s and "[boosting: gbdt]" in s


def test_master_seed_settles_the_other_random_streams():
    import numpy as np
    import lightgbm_amd as lgb
    rng = np.random.RandomState(0)
    X = rng.randn(3000, 6)
    y = X[:, 0] + 0.5 * rng.randn(3000)
    p = {"device_type": "cpu", "verbose": -1, "bagging_fraction": 0.5, "bagging_freq": 1, "feature_fraction": 0.5}
    fit = lambda **kw: lgb.train({**p, **kw}, lgb.Dataset(X, label=y), 5).predict(X)
    assert np.array_equal(fit(seed=1), fit(seed=1))
    assert not np.array_equal(fit(seed=1), fit(seed=2))                       # bag and feature subsets follow the seed
    assert np.array_equal(fit(seed=1, bagging_seed=9, feature_fraction_seed=9),
                          fit(seed=2, bagging_seed=9, feature_fraction_seed=9))   # explicit stream seeds win


def test_upstream_parameters_without_a_choice_here_are_accepted():
    """Switches between implementations this engine has one of (histogram orientation, OpenCL platform, bundling,
    two-round loading, gradient quantisation bins) parse as known parameters and do not change the model; the ones
    that change results upstream (num_grad_quant_bins, monotone_constraints_method) say once that they do not here."""
    import warnings
    from lightgbm_amd import basic
    basic._no_effect_warned.clear()
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_regression
    X, y = make_regression(1500, 5, seed=1)
    base = {"device_type": "cpu", "verbose": -1, "num_leaves": 7}
    extra = {"force_col_wise": True, "histogram_pool_size": 512.0, "gpu_platform_id": 0, "num_gpu": 1,
             "two_round": True, "enable_bundle": False, "num_grad_quant_bins": 8, "stochastic_rounding": False,
             "monotone_constraints_method": "advanced", "precise_float_parser": True}
    a = lgb.train(base, lgb.Dataset(X, label=y), 3)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        b = lgb.train({**base, **extra}, lgb.Dataset(X, label=y), 3)
        lgb.train({**base, **extra}, lgb.Dataset(X, label=y), 1)
    said = [str(w.message) for w in caught]
    assert sum("num_grad_quant_bins has no effect" in m for m in said) == 1
    assert sum("served by the basic method" in m for m in said) == 1
    cut = lambda s: s[s.index("Tree=0"):s.index("end of trees")]
    assert cut(a.model_to_string()) == cut(b.model_to_string())
    assert "[monotone_constraints_method: advanced]" in b.model_to_string()
    with pytest.raises(lgb.LightGBMError):
        lgb.train({**base, "monotone_constraints_method": "sometimes"}, lgb.Dataset(X, label=y), 1)


def test_saved_feature_importance_type_and_early_stopping_min_delta():
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_regression
    X, y = make_regression(3000, 5, seed=2)
    p = {"device_type": "cpu", "verbose": -1, "num_leaves": 7}
    b = lgb.train({**p, "saved_feature_importance_type": 1}, lgb.Dataset(X, label=y), 5)
    block = b.model_to_string().split("feature_importances:\n")[1].split("\n\n")[0]
    assert any("." in line.split("=")[1] for line in block.splitlines())            # gains, not counts
    block0 = b.model_to_string(importance_type="split").split("feature_importances:\n")[1].split("\n\n")[0]
    assert all(line.split("=")[1].isdigit() for line in block0.splitlines())
    # an improvement below min_delta does not count: training stops right after the patience runs out
    d = lgb.Dataset(X[:2000], label=y[:2000])
    v = d.create_valid(X[2000:], label=y[2000:])
    loose = lgb.train({**p, "learning_rate": 0.05}, d, 200, valid_sets=[v], early_stopping_rounds=5, verbose_eval=False)
    strict = lgb.train({**p, "learning_rate": 0.05, "early_stopping_min_delta": 1e9}, d, 200, valid_sets=[v],
                       early_stopping_rounds=5, verbose_eval=False)
    assert strict.best_iteration == 1 and strict.current_iteration() <= 7 < loose.current_iteration()


def test_pred_early_stop_keeps_confident_rows_short():
    """pred_early_stop: rows whose margin is wide after a check keep the score they have; a margin nobody reaches
    changes nothing (upstream: prediction_early_stop.cpp semantics - binary 2|score|, multiclass best - second)."""
    rng = np.random.RandomState(5)
    X = rng.randn(1500, 6)
    y = (X[:, 0] + 0.3 * rng.randn(1500) > 0).astype(float)
    b = lgb.train({"objective": "binary", "device_type": "cpu", "verbose": -1, "num_leaves": 15}, lgb.Dataset(X, label=y), 60)
    full = b.predict(X, raw_score=True)
    same = b.predict(X, raw_score=True, pred_early_stop=True, pred_early_stop_freq=7, pred_early_stop_margin=1e9)
    np.testing.assert_allclose(same, full, rtol=1e-12, atol=1e-12)
    early = b.predict(X, raw_score=True, pred_early_stop=True, pred_early_stop_freq=5, pred_early_stop_margin=3.0)
    stopped = np.abs(early - full) > 1e-9   # (summing the chunks' scores differs from one pass in the last bits)
    assert stopped.any() and not stopped.all()
    assert (2 * np.abs(early[stopped]) > 3.0).all()            # they stopped because the margin was reached
    assert ((early > 0) == (full > 0))[stopped].mean() > 0.99   # ... and on the side they end on
    # a stopped row holds the score of a whole number of checks
    at = {k: b.predict(X, raw_score=True, num_iteration=k) for k in range(5, 60, 5)}
    assert all(any(abs(early[i] - at[k][i]) < 1e-9 for k in at) for i in np.flatnonzero(stopped)[:50])
    prob = b.predict(X, pred_early_stop=True, pred_early_stop_freq=5, pred_early_stop_margin=3.0)
    np.testing.assert_allclose(prob, 1 / (1 + np.exp(-early)), rtol=1e-12)
    # multiclass, and the parameters taken from the booster
    y3 = np.digitize(X[:, 1], [-0.5, 0.5]).astype(float)
    p3 = {"objective": "multiclass", "num_class": 3, "device_type": "cpu", "verbose": -1, "num_leaves": 7,
          "pred_early_stop": True, "pred_early_stop_freq": 4, "pred_early_stop_margin": 2.0}
    b3 = lgb.train(p3, lgb.Dataset(X, label=y3), 40)
    e3 = b3.predict(X)
    f3 = b3.predict(X, pred_early_stop=False)
    assert e3.shape == f3.shape == (1500, 3)
    np.testing.assert_allclose(e3.sum(axis=1), 1.0, rtol=1e-12)
    assert (np.abs(e3 - f3) > 1e-9).any() and (e3.argmax(axis=1) == f3.argmax(axis=1)).mean() > 0.99
    # regression ignores it
    br = lgb.train({"objective": "regression", "device_type": "cpu", "verbose": -1}, lgb.Dataset(X, label=X[:, 0]), 10)
    np.testing.assert_array_equal(br.predict(X, pred_early_stop=True), br.predict(X))


def test_reg_sqrt_fits_root_labels_and_squares_predictions():
    """reg_sqrt (upstream: RegressionL2loss with sqrt_): the model is the one plain regression fits to
    sign(y) sqrt|y|, predictions are squared back, metrics are taken on the original scale, and the model text
    carries the flag."""
    rng = np.random.RandomState(9)
    X = rng.randn(2000, 5)
    y = (np.exp(X[:, 0]) + 0.1 * rng.rand(2000)) * np.where(X[:, 1] > 1.5, -1.0, 1.0)
    y = y.astype(np.float32).astype(np.float64)   # labels are held as floats: the root is taken of those
    base = {"device_type": "cpu", "verbose": -1, "num_leaves": 15, "min_data_in_leaf": 5}
    res = {}
    b = lgb.train({**base, "objective": "regression", "reg_sqrt": True, "metric": "l2"}, lgb.Dataset(X, label=y), 30,
                  valid_sets=[lgb.Dataset(X, label=y)], valid_names=["same"], evals_result=res, verbose_eval=False)
    yt = (np.sign(y) * np.sqrt(np.abs(y))).astype(np.float32)
    plain = lgb.train({**base, "objective": "regression"}, lgb.Dataset(X, label=yt), 30)
    raw = b.predict(X, raw_score=True)
    np.testing.assert_allclose(raw, plain.predict(X), rtol=1e-12, atol=1e-12)
    pred = b.predict(X)
    np.testing.assert_allclose(pred, np.sign(raw) * raw * raw, rtol=1e-12)
    np.testing.assert_allclose(res["same"]["l2"][-1], np.mean((pred - y) ** 2), rtol=1e-9)
    assert "objective=regression sqrt" in b.model_to_string()
    again = lgb.Booster(model_str=b.model_to_string())
    np.testing.assert_array_equal(again.predict(X), pred)
    assert np.mean((pred - y) ** 2) < 0.5 * np.var(y)


def test_forced_bin_bounds_are_kept(tmp_path):
    """forcedbins_filename: the listed upper bounds are among the feature's bin bounds (so a split can sit exactly
    there), the bin budget still holds, other features are binned as before."""
    import json
    rng = np.random.RandomState(3)
    X = rng.rand(4000, 3)
    y = (X[:, 0] > 0.3333).astype(float) + 0.01 * rng.randn(4000)
    path = tmp_path / "forced_bins.json"
    path.write_text(json.dumps([{"feature": 0, "bin_upper_bound": [0.3333, 0.5, 0.9]}, {"feature": 2, "bin_upper_bound": [0.25]}]))
    p = {"device_type": "cpu", "verbose": -1, "max_bin": 16, "min_data_in_bin": 1, "objective": "regression", "num_leaves": 4,
         "min_data_in_leaf": 1}
    plain = lgb.train(p, lgb.Dataset(X, label=y), 1)
    forced = lgb.train({**p, "forcedbins_filename": str(path)}, lgb.Dataset(X, label=y), 1)

    def thresholds(b, feature):
        out = set()
        def walk(n):
            if "split_feature" in n:
                if n["split_feature"] == feature:
                    out.add(n["threshold"])
                walk(n["left_child"]); walk(n["right_child"])
        for t in b.dump_model()["tree_info"]:
            walk(t["tree_structure"])
        return out
    assert 0.3333 in thresholds(forced, 0)            # the step sits on a forced bound ...
    assert 0.3333 not in thresholds(plain, 0)         # ... that quantile binning does not have
    d = lgb.Dataset(X, label=y, params={**p, "forcedbins_filename": str(path)}).construct()
    # bin budget: at most max_bin bins per feature (bounds from the model's thresholds over many trees)
    many = lgb.train({**p, "forcedbins_filename": str(path), "num_leaves": 31, "learning_rate": 0.5}, d, 40)
    assert len(thresholds(many, 0)) <= 15 and len(thresholds(many, 2)) <= 15
    assert {0.3333, 0.5, 0.9} & thresholds(many, 0) and len(thresholds(many, 1)) > 0
    with pytest.raises(Exception, match="forcedbins_filename"):
        bad = tmp_path / "bad.json"
        bad.write_text('[{"feature": 7, "bin_upper_bound": [1]}]')
        lgb.Dataset(X, label=y, params={**p, "forcedbins_filename": str(bad)}).construct()


def test_dataset_from_sequences_equals_the_matrix():
    """lgb.Sequence objects (rows served in batches) - one, or a list that is stacked - give the dataset the matrix gives."""
    class Rows(lgb.Sequence):
        def __init__(self, m, batch_size):
            self.m, self.batch_size, self.reads = m, batch_size, 0

        def __len__(self):
            return len(self.m)

        def __getitem__(self, idx):
            self.reads += 1
            return self.m[idx]

    rng = np.random.RandomState(2)
    X = rng.randn(3000, 4)
    y = X[:, 0] - X[:, 2] + 0.1 * rng.randn(3000)
    p = {"device_type": "cpu", "verbose": -1, "num_leaves": 15}
    want = lgb.train(p, lgb.Dataset(X, label=y), 10).model_to_string()
    one = Rows(X, 700)
    assert lgb.train(p, lgb.Dataset(one, label=y), 10).model_to_string() == want and one.reads == 5
    parts = [Rows(X[:1000], 256), Rows(X[1000:], 4096)]
    assert lgb.train(p, lgb.Dataset(parts, label=y), 10).model_to_string() == want
