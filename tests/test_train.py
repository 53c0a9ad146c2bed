"""End-to-end training on the CPU engine: quality, determinism, growth-order exactness, objectives, callbacks."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like, make_higgs_like, make_regression

CPU = {"device_type": "cpu", "verbose": -1}


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b)) ** 2)))


def trees_of(b):
    s = b.model_to_string()
    return s[s.index("Tree=0"):s.index("end of trees")]


def test_regression_quality_matches_sklearn_histgb():
    # diamonds-like regression must land within 2 % of sklearn's histogram GBDT
    from sklearn.ensemble import HistGradientBoostingRegressor
    X, y = make_diamonds_like(20000, seed=3)
    tr, te = slice(0, 16000), slice(16000, None)
    b = lgb.train({**CPU, "learning_rate": 0.1, "objective": "regression"}, lgb.Dataset(X[tr], label=y[tr]), 200)
    ours = rmse(b.predict(X[te]), y[te])
    sk = HistGradientBoostingRegressor(max_iter=200, learning_rate=0.1, max_leaf_nodes=31, min_samples_leaf=20,
                                       early_stopping=False).fit(X[tr], y[tr])
    theirs = rmse(sk.predict(X[te]), y[te])
    lin = np.linalg.lstsq(np.c_[X[tr], np.ones(16000)], y[tr], rcond=None)[0]
    linear = rmse(np.c_[X[te], np.ones(len(y[te]))] @ lin, y[te])
    assert ours < 1.02 * theirs
    assert ours < 0.8 * linear       # boosting beats the linear model (R:46 vs R:64)


def test_binary_quality_matches_sklearn_histgb():
    from sklearn.ensemble import HistGradientBoostingClassifier
    from sklearn.metrics import roc_auc_score
    from lightgbm_amd.utils.datasets import make_higgs_like
    X, y = make_higgs_like(30000, seed=5, device="numpy")
    tr, te = slice(0, 24000), slice(24000, None)
    ev = {}
    d = lgb.Dataset(X[tr], label=y[tr])
    v = d.create_valid(X[te], label=y[te])
    b = lgb.train({**CPU, "objective": "binary", "metric": ["auc", "binary_logloss"], "num_leaves": 31}, d, 100,
                  valid_sets=[v], evals_result=ev)
    p = b.predict(X[te])
    ours = roc_auc_score(y[te], p)
    assert abs(ev["valid_0"]["auc"][-1] - ours) < 1e-9          # our AUC metric == sklearn's on the same scores
    sk = HistGradientBoostingClassifier(max_iter=100, learning_rate=0.1, max_leaf_nodes=31, min_samples_leaf=20,
                                        early_stopping=False).fit(X[tr], y[tr])
    theirs = roc_auc_score(y[te], sk.predict_proba(X[te])[:, 1])
    assert ours > theirs - 0.005
    ll = -np.mean(y[te] * np.log(p) + (1 - y[te]) * np.log(1 - p))
    assert abs(ev["valid_0"]["binary_logloss"][-1] - ll) < 1e-9


def test_training_is_deterministic_and_thread_count_independent():
    X, y = make_regression(5000, 10, seed=1)
    a = lgb.train({**CPU, "num_threads": 1}, lgb.Dataset(X, label=y), 15)
    b = lgb.train({**CPU, "num_threads": 4}, lgb.Dataset(X, label=y), 15)
    assert trees_of(a) == trees_of(b)


@pytest.mark.parametrize("leaves,extra", [(31, {}), (63, {"min_data_in_leaf": 5}), (15, {"max_depth": 4}),
                                          (127, {"lambda_l2": 1.0, "min_gain_to_split": 0.01})])
def test_batched_growth_equals_strict_best_first(leaves, extra):
    # max_batch_leaves=1 splits one leaf per round = textbook best-first; the default batches every
    # frontier leaf that can still be reached.  Both must give the same trees.
    X, y = make_regression(6000, 8, seed=2)
    p = {**CPU, "num_leaves": leaves, **extra}
    strict = lgb.train({**p, "max_batch_leaves": 1}, lgb.Dataset(X, label=y), 8, keep_training_booster=True)
    batched = lgb.train(p, lgb.Dataset(X, label=y), 8, keep_training_booster=True)
    assert trees_of(strict) == trees_of(batched)
    assert batched.grow_stats()["rounds"] < strict.grow_stats()["rounds"]
    assert batched.grow_stats()["used_splits"] == strict.grow_stats()["used_splits"]


@pytest.mark.parametrize("flight", ["1", "3"])
def test_certain_splits_ahead_of_time_do_not_change_the_trees(flight, monkeypatch):
    """Leaves whose turn is certain (the budget outlasts everything that could be chosen ahead of them, given
    min_sum_hessian_in_leaf / min_data_in_leaf) are split before the replay reaches them: fewer rounds, same trees -
    with one round at a time and with several in flight (rounds dealt out over the free places)."""
    X, y = make_higgs_like(20000, seed=3, device="numpy")
    p = {**CPU, "objective": "binary", "num_leaves": 63, "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 20.0}
    monkeypatch.setenv("LGA_HOST_IN_FLIGHT", flight)
    monkeypatch.setenv("LGA_OWN_ROUND_ROWS", "3000")
    strict = lgb.train({**p, "max_batch_leaves": 1}, lgb.Dataset(X, label=y), 6)
    monkeypatch.setenv("LGA_CERTAIN_SPLITS", "0")
    estimates_only = lgb.train(p, lgb.Dataset(X, label=y), 6, keep_training_booster=True)
    monkeypatch.setenv("LGA_CERTAIN_SPLITS", "1")
    with_certain = lgb.train(p, lgb.Dataset(X, label=y), 6, keep_training_booster=True)
    assert trees_of(strict) == trees_of(estimates_only) == trees_of(with_certain)
    assert with_certain.grow_stats()["used_splits"] == estimates_only.grow_stats()["used_splits"]


def test_num_leaves_and_min_data_constraints_hold():
    X, y = make_regression(4000, 6, seed=4)
    b = lgb.train({**CPU, "num_leaves": 20, "min_data_in_leaf": 100}, lgb.Dataset(X, label=y), 5)
    for t in b.dump_model()["tree_info"]:
        assert t["num_leaves"] <= 20
        counts = []

        def walk(n):
            if "leaf_index" in n:
                counts.append(n["leaf_count"])
            else:
                walk(n["left_child"]); walk(n["right_child"])
        walk(t["tree_structure"])
        assert min(counts) >= 100 and sum(counts) == 4000


def test_leaf_values_are_newton_steps():
    # one tree, lr 1, L2: each leaf value = mean residual of its rows (+ init score)
    X, y = make_regression(3000, 5, seed=6)
    b = lgb.train({**CPU, "num_leaves": 8, "learning_rate": 1.0, "min_data_in_leaf": 50}, lgb.Dataset(X, label=y), 1)
    leaf = b.predict(X, pred_leaf=True)[:, 0]
    pred = b.predict(X)
    for l in np.unique(leaf):
        m = leaf == l
        # leaf values come from 36-bit sums of the float gradients (the 16-bit quantisation only places the splits);
        # what is left is the float32 rounding of the gradients themselves
        assert abs(pred[m][0] - y[m].mean()) < 1e-6 * max(1.0, abs(y[m].mean()))
    # without the renewal the values carry the 16-bit quantisation of the gradients
    q = lgb.train({**CPU, "num_leaves": 8, "learning_rate": 1.0, "min_data_in_leaf": 50, "quant_train_renew_leaf": False},
                  lgb.Dataset(X, label=y), 1)
    assert trees_of(q).split("leaf_value")[0] == trees_of(b).split("leaf_value")[0]   # same structure
    err = max(abs(q.predict(X)[leaf == l][0] - y[leaf == l].mean()) for l in np.unique(leaf))
    assert 1e-6 < err < 2e-3


def test_weights_and_init_score():
    X, y = make_regression(3000, 5, seed=7)
    w = np.where(X[:, 0] > 0, 2.0, 0.5)
    b = lgb.train({**CPU, "num_leaves": 8}, lgb.Dataset(X, label=y, weight=w), 5)
    b0 = lgb.train({**CPU, "num_leaves": 8}, lgb.Dataset(X, label=y), 5)
    assert trees_of(b) != trees_of(b0)
    # init_score: training on residuals of a constant offset gives the same trees as boost_from_average off
    d = lgb.Dataset(X, label=y, init_score=np.full(len(y), 3.0))
    bi = lgb.train({**CPU, "num_leaves": 8}, d, 3)
    br = lgb.train({**CPU, "num_leaves": 8, "boost_from_average": False}, lgb.Dataset(X, label=y - 3.0), 3)
    np.testing.assert_allclose(bi.predict(X), br.predict(X), atol=1e-5)   # labels are stored as float32


def test_early_stopping_and_best_iteration():
    X, y = make_regression(4000, 8, seed=8, noise=1.0)
    d = lgb.Dataset(X[:3000], label=y[:3000])
    v = d.create_valid(X[3000:], label=y[3000:])
    ev = {}
    b = lgb.train({**CPU, "learning_rate": 0.3, "metric": "rmse"}, d, 500, valid_sets=[v], early_stopping_rounds=5,
                  evals_result=ev)
    hist = ev["valid_0"]["rmse"]
    assert 0 < b.best_iteration < 500
    assert b.best_iteration == int(np.argmin(hist)) + 1
    assert len(hist) == b.best_iteration + 5
    np.testing.assert_allclose(b.best_score["valid_0"]["rmse"], min(hist))
    # predict uses best_iteration by default
    np.testing.assert_array_equal(b.predict(X[3000:]), b.predict(X[3000:], num_iteration=b.best_iteration))


def test_staged_prediction_and_rollback():
    X, y = make_regression(2000, 5, seed=9)
    b = lgb.train(CPU, lgb.Dataset(X, label=y), 20, keep_training_booster=True)
    p10 = b.predict(X, num_iteration=10)
    p20 = b.predict(X)
    assert rmse(p20, y) < rmse(p10, y) < rmse(b.predict(X, num_iteration=1), y)   # NBB:136 staged prediction
    np.testing.assert_allclose(b.predict(X, start_iteration=10, num_iteration=10, raw_score=True) + p10, p20, atol=1e-9)
    before = b._inner_predict(-1, raw=True)
    np.testing.assert_allclose(before, p20, atol=1e-9)
    b.rollback_one_iter()
    assert b.num_trees() == 19
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X), atol=1e-9)


def test_custom_objective_and_metric():
    X, y = make_regression(3000, 6, seed=10)

    def l2(preds, data):
        return preds - data.get_label(), np.ones_like(preds)

    def mae(preds, data):
        return "mae", float(np.mean(np.abs(preds - data.get_label()))), False
    d = lgb.Dataset(X, label=y)
    ev = {}
    b = lgb.train({**CPU, "boost_from_average": False}, d, 10, fobj=l2, feval=mae, valid_sets=[d], evals_result=ev)
    ref = lgb.train({**CPU, "boost_from_average": False}, lgb.Dataset(X, label=y), 10)
    np.testing.assert_allclose(b.predict(X), ref.predict(X), atol=1e-9)
    assert "mae" in ev["training"] and ev["training"]["mae"][-1] < ev["training"]["mae"][0]


def test_continue_training_from_init_model(tmp_path):
    X, y = make_regression(3000, 6, seed=11)
    d = lgb.Dataset(X, label=y, free_raw_data=False)
    b1 = lgb.train(CPU, d, 10)
    path = str(tmp_path / "m.txt")
    b1.save_model(path)
    b2 = lgb.train(CPU, lgb.Dataset(X, label=y, free_raw_data=False), 10, init_model=path)
    assert b2.num_trees() == 20
    assert rmse(b2.predict(X), y) < rmse(b1.predict(X), y)
    full = lgb.train(CPU, lgb.Dataset(X, label=y), 20)
    assert abs(rmse(b2.predict(X), y) - rmse(full.predict(X), y)) < 0.05 * rmse(full.predict(X), y)


def test_other_objectives_train():
    rng = np.random.RandomState(12)
    X = rng.randn(3000, 5)
    y = np.exp(0.5 * X[:, 0]) + 0.1 * rng.rand(3000)
    for obj in ("regression_l1", "huber", "fair", "poisson"):
        b = lgb.train({**CPU, "objective": obj, "num_leaves": 15}, lgb.Dataset(X, label=y), 30)
        assert np.corrcoef(b.predict(X), y)[0, 1] > 0.8, obj
    yb = (X[:, 0] + X[:, 1] > 0).astype(float)
    b = lgb.train({**CPU, "objective": "cross_entropy", "num_leaves": 15}, lgb.Dataset(X, label=yb), 30)
    assert np.mean((b.predict(X) > 0.5) == yb) > 0.95


def test_extra_trees_draws_one_threshold_per_feature_and_node():
    """extra_trees: a node only considers one random threshold per feature, so the trees differ from the greedy
    ones, depend on extra_seed, are reproducible, identical under one-leaf-at-a-time growth - and still learn."""
    X, y = make_regression(6000, 8, seed=17)
    d = lgb.Dataset(X, label=y, params=CPU)
    p = {**CPU, "num_leaves": 31, "learning_rate": 0.2}
    greedy = lgb.train(p, d, 30)
    et = lgb.train({**p, "extra_trees": True}, d, 30)
    assert trees_of(et) != trees_of(greedy)
    assert trees_of(et) == trees_of(lgb.train({**p, "extra_trees": True}, d, 30))
    assert trees_of(et) != trees_of(lgb.train({**p, "extra_trees": True, "extra_seed": 99}, d, 30))
    assert trees_of(lgb.train({**p, "extra_trees": True, "max_batch_leaves": 1}, d, 8)) == \
        trees_of(lgb.train({**p, "extra_trees": True}, d, 8))
    r_greedy, r_et = rmse(greedy.predict(X), y), rmse(et.predict(X), y)
    assert r_greedy < r_et < 2.5 * r_greedy          # randomised thresholds fit the training data less tightly
    # per node, the chosen gain can never beat the greedy root split
    g0 = greedy.dump_model()["tree_info"][0]["tree_structure"]["split_gain"]
    e0 = et.dump_model()["tree_info"][0]["tree_structure"]["split_gain"]
    assert e0 <= g0 + 1e-9


def test_monotone_constraints_hold_everywhere():
    """+1 / -1 constraints: along a constrained feature, with all other features held fixed, the prediction
    never moves the wrong way - checked on sweeps through many random base points; the unconstrained model on
    the same noisy data does violate it."""
    rng = np.random.RandomState(41)
    n = 6000
    X = rng.rand(n, 4) * 4 - 2
    y = 1.5 * X[:, 0] + np.sin(4 * X[:, 0]) - X[:, 1] ** 3 / 3 + np.cos(3 * X[:, 2]) + 0.5 * rng.randn(n)
    d = lgb.Dataset(X, label=y, params=CPU)
    p = {**CPU, "num_leaves": 31, "min_data_in_leaf": 10, "learning_rate": 0.2}
    free = lgb.train(p, d, 40)
    mono = lgb.train({**p, "monotone_constraints": [1, -1, 0, 0]}, d, 40)

    def violations(b, feature, sign):
        grid = np.linspace(-2, 2, 60)
        worst = 0.0
        for base in X[:80]:
            rows = np.tile(base, (len(grid), 1))
            rows[:, feature] = grid
            step = np.diff(b.predict(rows)) * sign
            worst = min(worst, step.min())
        return worst
    assert violations(mono, 0, +1) >= -1e-12 and violations(mono, 1, -1) >= -1e-12
    assert violations(free, 0, +1) < -0.05          # sin(4x) + noise: the free model wiggles
    assert rmse(mono.predict(X), y) < 1.5 * rmse(free.predict(X), y)
    assert "[monotone_constraints: 1,-1,0,0]" in mono.model_to_string()
    # deterministic batching: speculative growth picks the same splits as one leaf at a time
    one = lgb.train({**p, "monotone_constraints": [1, -1, 0, 0], "max_batch_leaves": 1}, d, 10)
    many = lgb.train({**p, "monotone_constraints": [1, -1, 0, 0]}, d, 10)
    assert trees_of(one) == trees_of(many)
    with pytest.raises(Exception, match="monotone_constraints"):
        lgb.train({**p, "monotone_constraints": [1, 0]}, d, 2)
    with pytest.raises(Exception):
        lgb.train({**p, "monotone_constraints": [2, 0, 0, 0]}, d, 2)


def test_dart_rescales_muted_trees_and_keeps_scores_and_model_in_step():
    X, y = make_regression(5000, 8, seed=33)
    d = lgb.Dataset(X[:4000], label=y[:4000], params=CPU)
    v = d.create_valid(X[4000:], label=y[4000:])
    p = {**CPU, "boosting": "dart", "num_leaves": 15, "learning_rate": 0.3, "drop_rate": 0.3, "skip_drop": 0.2,
         "metric": "l2"}
    ev = {}
    b = lgb.train(p, d, 40, valid_sets=[d, v], valid_names=["train", "valid"], evals_result=ev, keep_training_booster=True)
    assert b.num_trees() == 40
    # the scores the engine carried along == what the saved model predicts, on both datasets
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X[:4000], raw_score=True), atol=1e-9)
    np.testing.assert_allclose(b._inner_predict(0, raw=True), b.predict(X[4000:], raw_score=True), atol=1e-9)
    assert abs(ev["valid"]["l2"][-1] - np.mean((b.predict(X[4000:]) - y[4000:].astype(np.float32)) ** 2)) < 1e-9
    # muting happened: the stored shrinkage of early trees fell below the learning rate, and it learns
    shr = [float(ln.split("=")[1]) for ln in b.model_to_string().splitlines() if ln.startswith("shrinkage=")]
    assert min(shr) < 0.3 * 0.9 and max(shr) <= 0.3 + 1e-12
    assert ev["valid"]["l2"][-1] < 0.6 * ev["valid"]["l2"][0]     # muting 30 % of the trees slows learning down
    # deterministic; other seeds mute other trees; xgboost mode and uniform drop run too
    again = lgb.train(p, d, 40)
    assert trees_of(again) == trees_of(b)
    assert trees_of(lgb.train({**p, "drop_seed": 9}, d, 40)) != trees_of(b)
    for extra in ({"xgboost_dart_mode": True}, {"uniform_drop": True, "max_drop": 3}):
        c = lgb.train({**p, **extra}, d, 25, keep_training_booster=True)
        np.testing.assert_allclose(c._inner_predict(-1, raw=True), c.predict(X[:4000], raw_score=True), atol=1e-9)
    with pytest.raises(Exception, match="rolled back"):
        b.rollback_one_iter()


def test_goss_keeps_the_large_gradients_and_learns_almost_as_well():
    """Gradient-based one-side sampling: after the warm-up (1 / learning_rate iterations on all rows) every
    tree is grown on ~top_rate + other_rate of the rows, the heavy-gradient ones always among them."""
    X, y = make_regression(8000, 8, seed=31)
    d = lgb.Dataset(X, label=y, params=CPU)
    base = {**CPU, "num_leaves": 31, "learning_rate": 0.2}
    full = lgb.train(base, d, 60)
    goss = lgb.train({**base, "data_sample_strategy": "goss", "top_rate": 0.2, "other_rate": 0.1}, d, 60)
    old_name = lgb.train({**base, "boosting": "goss", "top_rate": 0.2, "other_rate": 0.1}, d, 60)
    assert trees_of(goss) == trees_of(old_name)
    # the first 1 / 0.2 = 5 trees see every row and equal plain boosting; later roots hold ~30 % of the rows
    tf, tg = full.dump_model()["tree_info"], goss.dump_model()["tree_info"]
    for i in range(5):
        assert tf[i] == tg[i]
    counts = [t["tree_structure"]["internal_count"] for t in tg[5:]]
    assert all(0.27 * 8000 < c < 0.34 * 8000 for c in counts), counts[:5]
    assert rmse(goss.predict(X), y) < 1.25 * rmse(full.predict(X), y)
    again = lgb.train({**base, "data_sample_strategy": "goss", "top_rate": 0.2, "other_rate": 0.1}, d, 60)
    assert trees_of(goss) == trees_of(again)
    for bad in ({"top_rate": 0.8, "other_rate": 0.5}, {"boosting": "rf", "bagging_fraction": 0.5, "bagging_freq": 1}):
        with pytest.raises(Exception):
            lgb.train({**base, "data_sample_strategy": "goss", **bad}, d, 2)


def test_l1_and_quantile_leaves_are_residual_percentiles():
    """One tree, learning rate 1, scores start at 0: a leaf's value is the percentile of the labels it holds."""
    X, y = make_regression(5000, 4, seed=8)
    y32 = y.astype(np.float32)
    for obj, alpha in (("regression_l1", 0.5), ("quantile", 0.8)):
        b = lgb.train({**CPU, "objective": obj, "alpha": alpha, "num_leaves": 8, "learning_rate": 1.0,
                       "boost_from_average": False}, lgb.Dataset(X, label=y, params=CPU), 1)
        leaf = b.predict(X, pred_leaf=True)[:, 0]
        for l in np.unique(leaf):
            v = np.sort(y32[leaf == l])
            n = len(v)
            pos = int((1 - alpha) * n)
            bias = (1 - alpha) * n - pos
            want = float(v[n - pos]) - (float(v[n - pos]) - float(v[n - 1 - pos])) * bias
            assert abs(b.get_leaf_output(0, int(l)) - want) < 1e-12
    # weighted: the weighted median splits the leaf's weight in half
    w = np.where(X[:, 1] > 0, 3.0, 1.0)
    b = lgb.train({**CPU, "objective": "regression_l1", "num_leaves": 6, "learning_rate": 1.0, "boost_from_average": False},
                  lgb.Dataset(X, label=y, weight=w, params=CPU), 1)
    leaf = b.predict(X, pred_leaf=True)[:, 0]
    for l in np.unique(leaf):
        m = leaf == l
        v = b.get_leaf_output(0, int(l))
        below = w[m][y32[m] < v].sum() / w[m].sum()
        assert 0.45 < below < 0.55


def test_quantile_mape_gamma_tweedie_objectives():
    """Quantile regression hits its coverage level; the log-link families (gamma, tweedie) fit a positive
    target; every objective's default metric is its own loss and it falls during training."""
    rng = np.random.RandomState(21)
    X = rng.randn(6000, 4)
    y = 2.0 + X[:, 0] + (0.5 + 0.5 * (X[:, 1] > 0)) * rng.randn(6000)
    d = lgb.Dataset(X, label=y, params=CPU)
    for alpha in (0.1, 0.9):
        ev = {}
        b = lgb.train({**CPU, "objective": "quantile", "alpha": alpha, "num_leaves": 15, "learning_rate": 0.2}, d, 80,
                      valid_sets=[d], evals_result=ev)
        assert abs(np.mean(y <= b.predict(X)) - alpha) < 0.04
        curve = ev["training"]["quantile"]
        assert curve[-1] < curve[0]
        u = y - b.predict(X)
        assert abs(curve[-1] - np.mean(np.where(u >= 0, alpha * u, (alpha - 1) * u))) < 1e-9
    ypos = np.exp(0.3 * X[:, 0] - 0.2 * X[:, 2]) * rng.gamma(4.0, 0.25, size=6000)
    dpos = lgb.Dataset(X, label=ypos, params=CPU)
    for obj, metric in (("gamma", "gamma"), ("tweedie", "tweedie"), ("mape", "mape")):
        ev = {}
        b = lgb.train({**CPU, "objective": obj, "num_leaves": 15}, dpos, 60, valid_sets=[dpos], evals_result=ev)
        assert list(ev["training"]) == [metric]
        assert ev["training"][metric][-1] < ev["training"][metric][0]
        p = b.predict(X)
        assert np.corrcoef(p, np.exp(0.3 * X[:, 0] - 0.2 * X[:, 2]))[0, 1] > 0.9, obj
        if obj != "mape":
            assert p.min() > 0 and np.allclose(p, np.exp(b.predict(X, raw_score=True)))
            # the model file names the objective, so a reloaded booster applies the log link as well
            np.testing.assert_array_equal(lgb.Booster(model_str=b.model_to_string()).predict(X), p)
    ev = {}
    lgb.train({**CPU, "objective": "gamma", "metric": ["gamma_deviance", "mape"]}, dpos, 20, valid_sets=[dpos],
              evals_result=ev)
    assert ev["training"]["gamma_deviance"][-1] < ev["training"]["gamma_deviance"][0]
    with pytest.raises(Exception):
        lgb.train({**CPU, "objective": "tweedie", "tweedie_variance_power": 2.5}, dpos, 2)


def test_feature_fraction_and_bagging_change_trees_but_stay_deterministic():
    X, y = make_regression(4000, 10, seed=13)
    p = {**CPU, "feature_fraction": 0.6, "bagging_fraction": 0.7, "bagging_freq": 2}
    a = lgb.train(p, lgb.Dataset(X, label=y), 10)
    b = lgb.train(p, lgb.Dataset(X, label=y), 10)
    c = lgb.train({**p, "bagging_seed": 11}, lgb.Dataset(X, label=y), 10)
    plain = lgb.train(CPU, lgb.Dataset(X, label=y), 10)
    assert trees_of(a) == trees_of(b) != trees_of(c)
    assert trees_of(a) != trees_of(plain)
    assert rmse(a.predict(X), y) < 1.6 * rmse(plain.predict(X), y)


def test_pandas_and_names():
    import pandas as pd
    X, y = make_regression(1000, 4, seed=14)
    df = pd.DataFrame(X, columns=["a", "b", "c d", "e"])
    b = lgb.train({**CPU, "num_leaves": 7}, lgb.Dataset(df, label=pd.Series(y)), 3)
    assert "feature_names=a b c_d e" in b.model_to_string()
    np.testing.assert_array_equal(b.predict(df), b.predict(X))


def test_snapshots_allow_resuming_a_killed_run(tmp_path):
    """snapshot_freq writes '<output_model>.snapshot_iter_k'; init_model continues from it to the same model."""
    X, y = make_regression(3000, 6, seed=8)
    out = str(tmp_path / "model.txt")
    p = {**CPU, "num_leaves": 15, "snapshot_freq": 4, "output_model": out}
    full = lgb.train(p, lgb.Dataset(X, label=y, params=CPU, free_raw_data=False), 10)
    import os
    assert os.path.exists(out + ".snapshot_iter_4") and os.path.exists(out + ".snapshot_iter_8")
    assert not os.path.exists(out + ".snapshot_iter_10")
    part = lgb.Booster(model_file=out + ".snapshot_iter_8")
    assert part.num_trees() == 8
    resumed = lgb.train({**CPU, "num_leaves": 15}, lgb.Dataset(X, label=y, params=CPU, free_raw_data=False), 2,
                        init_model=out + ".snapshot_iter_8")
    assert resumed.num_trees() == 10
    np.testing.assert_allclose(resumed.predict(X), full.predict(X), rtol=1e-9, atol=1e-9)


def test_zero_as_missing_sends_zeros_with_the_nans():
    rng = np.random.RandomState(0)
    X = rng.randn(6000, 3)
    absent = rng.rand(6000) < 0.3
    X[absent, 0] = 0.0                       # "not recorded" is written as 0
    X[rng.rand(6000) < 0.05, 0] = np.nan
    y = np.where(absent | np.isnan(X[:, 0]), 2.0, X[:, 0]) + 0.1 * rng.randn(6000)
    p = {**CPU, "num_leaves": 15, "zero_as_missing": True}
    d = lgb.Dataset(X, label=y, params=p)
    b = lgb.train(p, d, 30, keep_training_booster=True)
    assert d.bin_info(0)["missing_bin"] == d.bin_info(0)["num_bin"] - 1
    df = b.trees_to_dataframe()
    assert set(df[df.split_feature == "Column_0"]["missing_type"]) == {"Zero"}
    assert float(np.mean((b.predict(X) - y) ** 2)) < 0.03
    # zeros and NaNs take the same branch everywhere; the binned training scores agree with the raw-value predictor
    probe = np.tile(X[:1], (2, 1))
    probe[:, 0] = [0.0, np.nan]
    assert b.predict(probe)[0] == b.predict(probe)[1]
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True), atol=1e-9)
    from lightgbm_amd.models import TreeEnsemble
    np.testing.assert_allclose(TreeEnsemble.from_string(b.model_to_string()).predict(X), b.predict(X), atol=1e-12)


def _internal_nodes(node):
    if "split_feature" in node:
        yield node
        yield from _internal_nodes(node["left_child"])
        yield from _internal_nodes(node["right_child"])


def test_feature_fraction_bynode_samples_per_node_and_is_order_independent():
    """Per-node feature subsets: named by tree position, so batched / speculative growth and the textbook
    one-leaf-at-a-time growth pick the same splits; the model differs from the unsampled one and still learns."""
    X, y = make_regression(6000, 12, seed=14)
    ds = lgb.Dataset(X, label=y, params=CPU)
    p = {**CPU, "num_leaves": 31, "feature_fraction_bynode": 0.4, "min_data_in_leaf": 10}
    a = lgb.train(p, ds, 12)
    b = lgb.train({**p, "max_batch_leaves": 1}, ds, 12)
    assert trees_of(a) == trees_of(b)
    full = lgb.train({**CPU, "num_leaves": 31, "min_data_in_leaf": 10}, ds, 12)
    assert trees_of(a) != trees_of(full)
    rmse = lambda m: float(np.sqrt(np.mean((m.predict(X) - y) ** 2)))
    assert rmse(a) < 0.5 * float(np.std(y))
    # different trees draw different subsets
    used = [set(t["split_feature"] for t in _internal_nodes(tr["tree_structure"])) for tr in a.dump_model()["tree_info"][:4]]
    assert len({frozenset(u) for u in used}) > 1


def _branches(booster):
    """Feature sets of all root-to-leaf branches of a model."""
    out = []

    def walk(node, used):
        if "split_feature" in node:
            u = used | {node["split_feature"]}
            walk(node["left_child"], u)
            walk(node["right_child"], u)
        else:
            out.append(used)
    for t in booster.dump_model()["tree_info"]:
        walk(t["tree_structure"], frozenset())
    return out


def test_interaction_constraints_keep_every_branch_inside_one_group():
    rng = np.random.RandomState(0)
    X = rng.randn(8000, 6)
    y = X[:, 0] * X[:, 1] + X[:, 2] * X[:, 3] + 0.5 * X[:, 4] + X[:, 5] + 0.1 * rng.randn(8000)
    base = {**CPU, "num_leaves": 31, "min_data_in_leaf": 20}
    b = lgb.train({**base, "interaction_constraints": [[0, 1], [2, 3, 4]]}, lgb.Dataset(X, label=y), 25)
    paths = _branches(b)
    assert all(p <= {0, 1} or p <= {2, 3, 4} for p in paths)
    assert set().union(*paths) == {0, 1, 2, 3, 4}            # column 5 is in no group: never used
    assert any(len(p) >= 2 for p in paths)
    # overlapping groups: a branch that has used the shared feature only may still go either way
    b2 = lgb.train({**base, "interaction_constraints": "[0,1,2],[2,3]"}, lgb.Dataset(X, label=y), 25)
    assert all(p <= {0, 1, 2} or p <= {2, 3} for p in _branches(b2))
    # the string survives the model file
    assert "[interaction_constraints: [0,1,2],[2,3]]" in b2.model_to_string()
    with pytest.raises(lgb.LightGBMError, match="does not exist"):
        lgb.train({**base, "interaction_constraints": [[0, 9]]}, lgb.Dataset(X, label=y), 2)
    with pytest.raises(lgb.LightGBMError, match="expected a list"):
        lgb.train({**base, "interaction_constraints": "0,1"}, lgb.Dataset(X, label=y), 2)


def test_path_smooth_shrinks_small_leaves_towards_their_parents():
    rng = np.random.RandomState(1)
    X = rng.randn(3000, 5)
    y = X[:, 0] + 3.0 * rng.randn(3000)                     # noisy: small leaves overfit
    Xv = rng.randn(3000, 5)
    yv = Xv[:, 0] + 3.0 * rng.randn(3000)
    base = {**CPU, "num_leaves": 63, "min_data_in_leaf": 2, "learning_rate": 0.3}
    plain = lgb.train(base, lgb.Dataset(X, label=y), 20)
    smooth = lgb.train({**base, "path_smooth": 30.0}, lgb.Dataset(X, label=y), 20)
    l2 = lambda m, A, t: float(np.mean((m.predict(A) - t) ** 2))
    assert l2(smooth, X, y) > l2(plain, X, y)               # fits the training noise less ...
    assert l2(smooth, Xv, yv) < l2(plain, Xv, yv)           # ... and generalises better
    # one tree by hand: a stump's children are w * raw + (1 - w) * root output, w = (n / s) / (n / s + 1)
    st = lgb.train({**CPU, "num_leaves": 2, "min_data_in_leaf": 2, "learning_rate": 1.0, "path_smooth": 30.0,
                    "boost_from_average": False}, lgb.Dataset(X, label=y), 1)
    t = st.dump_model()["tree_info"][0]["tree_structure"]
    thr, f = t["threshold"], t["split_feature"]
    left = X[:, f] <= thr
    root_out = y.mean()
    for child, rows in ((t["left_child"], left), (t["right_child"], ~left)):
        n = rows.sum()
        w = (n / 30.0) / (n / 30.0 + 1.0)
        assert child["leaf_value"] == pytest.approx(w * y[rows].mean() + (1 - w) * root_out, rel=2e-3)
    # with monotone constraints on top the ordering still holds
    m = lgb.train({**base, "path_smooth": 30.0, "monotone_constraints": [1, 0, 0, 0, 0]}, lgb.Dataset(X, label=y), 10)
    grid = np.tile(X[:1], (50, 1))
    grid[:, 0] = np.linspace(-3, 3, 50)
    assert np.all(np.diff(m.predict(grid)) >= -1e-12)


def test_average_precision_matches_sklearn():
    from sklearn.metrics import average_precision_score
    rng = np.random.RandomState(0)
    X = rng.randn(5000, 6)
    y = (X[:, 0] + rng.randn(5000) > 1.0).astype(float)
    w = 0.5 + rng.rand(5000)
    d = lgb.Dataset(X, label=y, weight=w)
    ev = {}
    b = lgb.train({**CPU, "objective": "binary", "metric": ["auc", "average_precision"], "num_leaves": 7}, d, 10,
                  valid_sets=[d], evals_result=ev)
    assert list(ev["training"]) == ["auc", "average_precision"]
    want = average_precision_score(y, b.predict(X, raw_score=True), sample_weight=w)    # few leaves: tied scores
    assert ev["training"]["average_precision"][-1] == pytest.approx(want, abs=1e-8)
    # maximised by early stopping
    b2 = lgb.train({**CPU, "objective": "binary", "metric": "average_precision", "learning_rate": 0.5}, d, 300,
                   valid_sets=[lgb.Dataset(X[:1000] + rng.randn(1000, 6), label=y[:1000], reference=d)], early_stopping_rounds=5)
    assert 0 < b2.best_iteration < 300


def test_scipy_sparse_input_is_accepted_for_training_and_prediction():
    sp = pytest.importorskip("scipy.sparse")
    rng = np.random.RandomState(0)
    X = rng.randn(2000, 10) * (rng.rand(2000, 10) < 0.2)
    y = X[:, 0] - X[:, 1] + 0.1 * rng.randn(2000)
    dense = lgb.train(CPU, lgb.Dataset(X, label=y), 5)
    for fmt in (sp.csr_matrix, sp.csc_matrix, sp.coo_matrix):
        S = fmt(X)
        b = lgb.train(CPU, lgb.Dataset(S, label=y), 5)
        assert b.model_to_string() == dense.model_to_string()
        np.testing.assert_array_equal(b.predict(S), dense.predict(X))


def test_balanced_bagging_keeps_different_shares_of_the_classes():
    rng = np.random.RandomState(0)
    X = rng.randn(20000, 5)
    y = (X[:, 0] + rng.randn(20000) > 1.6).astype(float)            # about 13 % positives
    base = {**CPU, "objective": "binary", "num_leaves": 15, "bagging_freq": 1, "boost_from_average": False}
    plain = lgb.train(base, lgb.Dataset(X, label=y), 20)
    bal = lgb.train({**base, "pos_bagging_fraction": 1.0, "neg_bagging_fraction": 0.15}, lgb.Dataset(X, label=y), 20)
    # the trees see about as many negatives as positives: predicted probabilities move up
    assert bal.predict(X).mean() > plain.predict(X).mean() + 0.1
    from sklearn.metrics import roc_auc_score
    assert roc_auc_score(y, bal.predict(X)) > 0.85
    # out-of-bag rows still get their scores: training scores equal predictions
    kept = lgb.Booster({**base, "pos_bagging_fraction": 0.8, "neg_bagging_fraction": 0.3}, lgb.Dataset(X, label=y))
    for _ in range(4):
        kept.update()
    np.testing.assert_allclose(kept._inner_predict(-1, raw=True), kept.predict(X, raw_score=True), atol=1e-9)
    with pytest.raises(lgb.LightGBMError, match="binary"):
        lgb.train({**CPU, "neg_bagging_fraction": 0.5, "bagging_freq": 1}, lgb.Dataset(X, label=y), 2)


def test_feature_contri_and_cost_effective_penalties():
    rng = np.random.RandomState(0)
    X = rng.randn(6000, 4)
    X[:, 1] = X[:, 0] + 0.05 * rng.randn(6000)               # column 1 is nearly a copy of column 0
    y = X[:, 0] + 0.5 * X[:, 2] + 0.1 * rng.randn(6000)
    base = {**CPU, "num_leaves": 15, "min_data_in_leaf": 20}
    used = lambda b: b.feature_importance().tolist()
    plain = lgb.train(base, lgb.Dataset(X, label=y), 10)
    assert used(plain)[0] > used(plain)[1]
    # a small factor on column 0's gains hands its splits to the copy; a zero switches a column off
    tilted = lgb.train({**base, "feature_contri": [0.05, 1.0, 1.0, 1.0]}, lgb.Dataset(X, label=y), 10)
    assert used(tilted)[1] > used(tilted)[0]
    off = lgb.train({**base, "feature_contri": [0.0, 1.0, 1.0, 1.0]}, lgb.Dataset(X, label=y), 10)
    assert used(off)[0] == 0 and used(off)[1] > 0
    # coupled cost: a feature is only bought when its gain pays for it - the noise column never is, and a
    # feature that was bought once is free afterwards
    gains = plain.feature_importance(importance_type="gain")
    costly = lgb.train({**base, "cegb_penalty_feature_coupled": [0.0, 0.0, 0.0, 10 * float(gains[3] + 1.0)]},
                       lgb.Dataset(X, label=y), 10)
    assert used(costly)[3] == 0 and used(costly)[2] > 0
    bought = lgb.train({**base, "cegb_penalty_feature_coupled": [0.0, 0.0, 0.3 * float(gains[2]) / 10, 0.0]}, lgb.Dataset(X, label=y), 10)
    assert used(bought)[2] > 0
    # a cost per row of every split leaf makes trees smaller
    nodes = lambda b: len(b.trees_to_dataframe())
    assert nodes(lgb.train({**base, "cegb_penalty_split": 0.05}, lgb.Dataset(X, label=y), 10)) < nodes(plain)
    with pytest.raises(lgb.LightGBMError, match="feature_contri"):
        lgb.train({**base, "feature_contri": [1.0, 1.0]}, lgb.Dataset(X, label=y), 2)
    with pytest.raises(lgb.LightGBMError, match="lazy"):
        lgb.train({**base, "cegb_penalty_feature_lazy": [1.0, 0, 0, 0]}, lgb.Dataset(X, label=y), 2)


def _heavy_tailed(n, seed):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, 8)
    y = X[:, 0] * 2 + np.sin(3 * X[:, 1]) + 0.5 * X[:, 2] * X[:, 3] + 0.1 * rng.randn(n)
    big = rng.rand(n) < 0.002          # a few targets ~100x the typical scale: they set the quantisation range
    y = np.where(big, y + 150.0 * np.sign(rng.randn(n)), y)
    return X, y


def test_heavy_tailed_regression_matches_full_precision_and_sklearn():
    """A few gradients 100x the typical size stretch the 16-bit quantisation range over the bulk of the rows.
    The quantised learner must stay within 1 % RMSE of the full-precision one (36-bit histograms) and of
    scikit-learn's histogram boosting on the bulk of held-out rows."""
    from sklearn.ensemble import HistGradientBoostingRegressor
    X, y = _heavy_tailed(40000, 3)
    tr, te = slice(0, 30000), slice(30000, None)
    bulk = np.abs(y[te]) < 20
    p = {**CPU, "num_leaves": 31, "learning_rate": 0.1}
    quant = lgb.train(p, lgb.Dataset(X[tr], label=y[tr]), 150)
    full = lgb.train({**p, "use_quantized_grad": False}, lgb.Dataset(X[tr], label=y[tr]), 150)
    sk = HistGradientBoostingRegressor(max_iter=150, learning_rate=0.1, max_leaf_nodes=31, min_samples_leaf=20,
                                       early_stopping=False).fit(X[tr], y[tr])
    rq, rf, rs = (rmse(m.predict(X[te])[bulk], y[te][bulk]) for m in (quant, full, sk))
    assert rq < 1.01 * rf and rq < 1.01 * rs, (rq, rf, rs)


def test_quantised_binary_auc_matches_full_precision():
    """Held-out AUC of the default learner (16-bit gradient histograms, renewed leaf values) against the
    full-precision one on a Higgs-like task: within 0.001."""
    from sklearn.metrics import roc_auc_score
    from lightgbm_amd.utils.datasets import make_higgs_like
    X, y = make_higgs_like(120000, seed=11, device="numpy")
    tr, te = slice(0, 100000), slice(100000, None)
    p = {**CPU, "objective": "binary", "num_leaves": 63, "learning_rate": 0.1, "min_sum_hessian_in_leaf": 10.0}
    quant = lgb.train(p, lgb.Dataset(X[tr], label=y[tr]), 120)
    full = lgb.train({**p, "use_quantized_grad": False}, lgb.Dataset(X[tr], label=y[tr]), 120)
    aq, af = roc_auc_score(y[te], quant.predict(X[te])), roc_auc_score(y[te], full.predict(X[te]))
    assert abs(aq - af) < 0.001, (aq, af)
    # the full-precision histograms really are finer: the two learners do not grow identical trees
    assert trees_of(quant) != trees_of(full)


def test_stochastic_rounding_is_seeded_unbiased_and_changes_the_16_bit_trees():
    """stochastic_rounding=true rounds the 16-bit gradients floor(s + u), u from the counter-based generator keyed by
    (seed, tree, row): reproducible for a seed, different between seeds and from round-to-nearest, and the model is as
    good; full-precision histograms do not round at that level, so the switch changes nothing there."""
    X, y = _heavy_tailed(20000, 5)
    p = {**CPU, "num_leaves": 31, "learning_rate": 0.1}
    ds = lambda: lgb.Dataset(X, label=y)
    near = lgb.train(p, ds(), 30)
    s1 = lgb.train({**p, "stochastic_rounding": True, "seed": 1}, ds(), 30)
    s1b = lgb.train({**p, "stochastic_rounding": True, "seed": 1}, ds(), 30)
    s2 = lgb.train({**p, "stochastic_rounding": True, "seed": 2}, ds(), 30)
    assert trees_of(s1) == trees_of(s1b)
    assert trees_of(s1) != trees_of(near) and trees_of(s1) != trees_of(s2)
    l2 = lambda b: float(np.mean((b.predict(X) - y) ** 2))
    assert abs(l2(s1) - l2(near)) < 0.02 * l2(near)
    f0 = lgb.train({**p, "use_quantized_grad": False}, ds(), 10)
    f1 = lgb.train({**p, "use_quantized_grad": False, "stochastic_rounding": True}, ds(), 10)
    assert trees_of(f0) == trees_of(f1)
    # gpu_use_dp is upstream's switch for full-precision device histograms
    f2 = lgb.train({**p, "gpu_use_dp": True}, ds(), 10)
    assert trees_of(f0) == trees_of(f2)


def test_forced_splits_start_every_tree(tmp_path):
    """forcedsplits_filename: the root splits on column 3 at 0.25, its left child on column 1 at -0.5, whatever the
    search would have chosen; growth below continues as usual (best-first over the remaining budget)."""
    import json
    X, y = make_regression(5000, 6, seed=9)
    path = tmp_path / "forced.json"
    path.write_text(json.dumps({"feature": 3, "threshold": 0.25, "left": {"feature": 1, "threshold": -0.5}, "right": {}}))
    p = {**CPU, "num_leaves": 15, "min_data_in_leaf": 20}
    free = lgb.train(p, lgb.Dataset(X, label=y), 3)
    forced = lgb.train({**p, "forcedsplits_filename": str(path)}, lgb.Dataset(X, label=y), 3)
    assert free.dump_model()["tree_info"][0]["tree_structure"]["split_feature"] != 3      # the search prefers another column
    for t in forced.dump_model()["tree_info"]:
        root = t["tree_structure"]
        assert root["split_feature"] == 3 and 0.25 <= root["threshold"] < 0.3
        left = root["left_child"]
        assert left["split_feature"] == 1 and -0.5 <= left["threshold"] < -0.45
        assert t["num_leaves"] == 15
    # rows really go where the model says: predictions of the text model equal the training scores
    np.testing.assert_allclose(forced.predict(X), lgb.Booster(model_str=forced.model_to_string()).predict(X), atol=1e-12)
    # the forced model cannot beat the free one on the training data, but must still learn
    l2 = lambda b: float(np.mean((b.predict(X) - y) ** 2))
    assert l2(free) <= l2(forced) < float(np.var(y))
    # a threshold beyond the data ends the forcing instead of making an empty child
    path.write_text(json.dumps({"feature": 3, "threshold": 1e9}))
    same = lgb.train({**p, "forcedsplits_filename": str(path)}, lgb.Dataset(X, label=y), 3)
    assert trees_of(same) == trees_of(free)
    path.write_text('{"feature": 3}')
    with pytest.raises(lgb.LightGBMError):
        lgb.train({**p, "forcedsplits_filename": str(path)}, lgb.Dataset(X, label=y), 1)
