"""sklearn-style estimators and the key=value command line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.sklearn import LGBMClassifier, LGBMRandomForestRegressor, LGBMRegressor
from lightgbm_amd.utils.datasets import make_curve_1d, make_regression

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_regressor_and_forest_follow_the_sklearn_protocol():
    X, y = make_regression(3000, 6, seed=0)
    m = LGBMRegressor(n_estimators=40, num_leaves=15, device_type="cpu").fit(X[:2400], y[:2400], eval_set=[(X[2400:], y[2400:])],
                                                                          eval_metric="rmse", early_stopping_rounds=10)
    assert m.score(X[2400:], y[2400:]) > 0.8 and m.feature_importances_.sum() > 0
    assert "valid_0" in m.evals_result_ and m.get_params()["num_leaves"] == 15
    xg, yg, xt, yt = make_curve_1d()
    rf = LGBMRandomForestRegressor(n_estimators=100, num_leaves=20, min_child_samples=5, random_state=345,
                                   device_type="cpu").fit(xt, yt)          # NBB:204
    assert np.sqrt(np.mean((rf.predict(xg) - yg) ** 2)) < 0.08


def test_classifier_probabilities_and_labels():
    rng = np.random.RandomState(1)
    X = rng.randn(2000, 5)
    y = np.where(X[:, 0] + X[:, 1] > 0, "yes", "no")
    c = LGBMClassifier(n_estimators=30, device_type="cpu").fit(X, y)
    proba = c.predict_proba(X)
    assert proba.shape == (2000, 2) and np.allclose(proba.sum(axis=1), 1)
    assert set(c.predict(X)) <= {"yes", "no"} and np.mean(c.predict(X) == y) > 0.95


def test_cli_train_predict_convert(tmp_path):
    X, y = make_regression(1500, 4, seed=2)
    data = str(tmp_path / "train.csv")
    np.savetxt(data, np.column_stack([y, X]), delimiter=",")
    model, pred, js = str(tmp_path / "m.txt"), str(tmp_path / "p.txt"), str(tmp_path / "m.json")
    env = {**os.environ, "LGA_FORCE_HOST": "1", "PYTHONPATH": ROOT}
    run = lambda *a: subprocess.run([sys.executable, "-m", "lightgbm_amd.cli", *a], cwd=ROOT, env=env, capture_output=True,
                                    text=True, timeout=300)
    r = run("task=train", f"data={data}", f"output_model={model}", "num_trees=20", "num_leaves=15", "verbose=-1",
            "objective=regression")
    assert r.returncode == 0, r.stderr
    r = run("task=predict", f"data={data}", f"input_model={model}", f"output_result={pred}")
    assert r.returncode == 0, r.stderr
    p = np.loadtxt(pred)
    assert np.corrcoef(p, y)[0, 1] > 0.9
    r = run("task=convert_model", f"input_model={model}", f"convert_model={js}")
    assert r.returncode == 0 and len(json.load(open(js))["tree_info"]) == 20


def test_cli_continues_a_multiclass_model(tmp_path):
    # input_model on task=train: 4 + 4 iterations of a 3-class model equal 8 in one go
    rng = np.random.RandomState(0)
    X = rng.randn(600, 5)
    y = np.digitize(X[:, 0] + 0.5 * X[:, 1], [-0.6, 0.6])
    data = str(tmp_path / "train.csv")
    np.savetxt(data, np.column_stack([y, X]), delimiter=",", fmt="%.6f")
    env = {**os.environ, "LGA_FORCE_HOST": "1", "PYTHONPATH": ROOT}
    common = ["task=train", f"data={data}", "objective=multiclass", "num_class=3", "device_type=cpu", "verbose=-1",
              "min_data_in_leaf=10", "num_leaves=7"]
    run = lambda *a: subprocess.run([sys.executable, "-m", "lightgbm_amd.cli", *common, *a], cwd=ROOT, env=env,
                                    capture_output=True, text=True, timeout=300)
    m4, m8, w8 = (str(tmp_path / n) for n in ("m4.txt", "m8.txt", "w8.txt"))
    for args in (["num_iterations=4", f"output_model={m4}"], ["num_iterations=4", f"input_model={m4}", f"output_model={m8}"],
                 ["num_iterations=8", f"output_model={w8}"]):
        r = run(*args)
        assert r.returncode == 0, r.stderr
    Xr = np.loadtxt(data, delimiter=",")[:, 1:]
    a, b = lgb.Booster(model_file=m8), lgb.Booster(model_file=w8)
    assert a.num_trees() == b.num_trees() == 24
    np.testing.assert_allclose(a.predict(Xr), b.predict(Xr), rtol=1e-6, atol=1e-9)


def test_estimators_work_inside_sklearn_grid_search():
    """"While already available in Python's sklearn" ("LightGBM R.ipynb", purpose cell): the estimators clone,
    take parameters and score like any scikit-learn regressor, so GridSearchCV drives them directly."""
    from sklearn.base import clone
    from sklearn.model_selection import GridSearchCV, cross_val_score
    from lightgbm_amd.utils.datasets import make_diamonds_like
    X, y = make_diamonds_like(3000, seed=3)
    est = LGBMRegressor(n_estimators=30, device_type="cpu", verbose=-1)
    twin = clone(est)
    assert twin.get_params()["n_estimators"] == 30 and twin is not est
    gs = GridSearchCV(est, {"learning_rate": [0.3, 0.03], "num_leaves": [7, 31]}, cv=3,
                      scoring="neg_root_mean_squared_error").fit(X, y)
    assert set(gs.best_params_) == {"learning_rate", "num_leaves"} and gs.best_params_["learning_rate"] == 0.3
    assert gs.best_estimator_.predict(X[:5]).shape == (5,)
    scores = cross_val_score(LGBMClassifier(n_estimators=10, device_type="cpu", verbose=-1), X, (y > np.median(y)).astype(int), cv=3)
    assert scores.mean() > 0.9


def test_classifier_class_weight():
    from lightgbm_amd.sklearn import LGBMClassifier
    rng = np.random.RandomState(0)
    X = rng.randn(4000, 5)
    y = np.where(X[:, 0] + rng.randn(4000) > 1.2, "rare", "common")
    plain = LGBMClassifier(n_estimators=15, device_type="cpu", verbose=-1).fit(X, y)
    balanced = LGBMClassifier(n_estimators=15, class_weight="balanced", device_type="cpu", verbose=-1).fit(X, y)
    by_hand = LGBMClassifier(n_estimators=15, class_weight={"rare": 3.0}, device_type="cpu", verbose=-1).fit(X, y)
    rate = lambda m: float((m.predict(X) == "rare").mean())
    assert rate(balanced) > rate(by_hand) > rate(plain)
    with pytest.raises(ValueError):
        LGBMClassifier(class_weight="heavy", device_type="cpu").fit(X, y)


def test_cli_libsvm_ranking_side_files_refit_and_binary_cache(tmp_path):
    """The application's file conventions: LibSVM rows, <data>.query / <data>.weight side files, continued
    training from input_model, the refit task and the binned-dataset cache."""
    rng = np.random.RandomState(0)
    sizes = rng.randint(5, 20, size=80)
    n = int(sizes.sum())
    X = np.round(rng.randn(n, 5), 3) * (rng.rand(n, 5) < 0.7)
    y = np.clip(np.round(X[:, 0] + 1.5), 0, 3)
    data = str(tmp_path / "rank.train")
    with open(data, "w") as f:
        for i in range(n):
            f.write(f"{int(y[i])} " + " ".join(f"{j}:{X[i, j]}" for j in range(5) if X[i, j] != 0) + "\n")
    np.savetxt(data + ".query", sizes, fmt="%d")
    np.savetxt(data + ".weight", 0.5 + rng.rand(n))
    env = {**os.environ, "LGA_FORCE_HOST": "1", "PYTHONPATH": ROOT}
    run = lambda *a: subprocess.run([sys.executable, "-m", "lightgbm_amd.cli", *a], cwd=ROOT, env=env, capture_output=True,
                                    text=True, timeout=300)
    m1, m2, m3, pred = (str(tmp_path / k) for k in ("m1.txt", "m2.txt", "m3.txt", "p.txt"))
    common = ("objective=lambdarank", "num_leaves=7", "min_data_in_leaf=5", "verbose=-1", "eval_at=3")
    r = run("task=train", f"data={data}", f"valid={data}", f"output_model={m1}", "num_trees=8", *common)
    assert r.returncode == 0, r.stderr
    assert "ndcg@3" in r.stdout
    r = run("task=train", f"data={data}", f"input_model={m1}", f"output_model={m2}", "num_trees=4", *common)
    assert r.returncode == 0, r.stderr
    import lightgbm_amd as lgb
    assert lgb.Booster(model_file=m2).num_trees() == 12
    r = run("task=predict", f"data={data}", f"input_model={m2}", f"output_result={pred}", "predict_leaf_index=true")
    assert r.returncode == 0 and np.loadtxt(pred).shape == (n, 12), r.stderr
    # regression model: refit on shifted labels, binned cache round trip
    reg = str(tmp_path / "reg.tsv")
    Xr = rng.randn(1500, 4)
    yr = Xr[:, 0] + 0.1 * rng.randn(1500)
    np.savetxt(reg, np.column_stack([yr, Xr]), delimiter="\t")
    assert run("task=train", f"data={reg}", f"output_model={m3}", "num_trees=15", "verbose=-1").returncode == 0
    shifted = str(tmp_path / "shift.tsv")
    np.savetxt(shifted, np.column_stack([yr + 3.0, Xr]), delimiter="\t")
    out = str(tmp_path / "refit.txt")
    r = run("task=refit", f"data={shifted}", f"input_model={m3}", f"output_model={out}", "refit_decay_rate=0.0")
    assert r.returncode == 0, r.stderr
    # 15 shrunken Newton steps take up 1 - 0.9^15 = 79 % of the shift
    assert abs(float(np.mean(lgb.Booster(model_file=out).predict(Xr) - yr)) - 3.0 * (1 - 0.9 ** 15)) < 0.05
    assert run("task=save_binary", f"data={reg}", "verbose=-1").returncode == 0
    r = run("task=train", f"data={reg}.bin", f"output_model={m1}", "num_trees=15", "verbose=-1")
    assert r.returncode == 0, r.stderr
    assert open(m1).read().split("parameters:")[0] == open(m3).read().split("parameters:")[0]


def test_dataset_from_a_text_file_path(tmp_path):
    import lightgbm_amd as lgb
    rng = np.random.RandomState(0)
    X = rng.randn(500, 4)
    y = X[:, 0] + 0.1 * rng.randn(500)
    path = str(tmp_path / "t.csv")
    np.savetxt(path, np.column_stack([X[:, :2], y, X[:, 2:]]), delimiter=",", header="a,b,target,c,d", comments="")
    np.savetxt(path + ".weight", np.full(500, 2.0))
    p = {"device_type": "cpu", "verbose": -1}
    from_file = lgb.train(p, lgb.Dataset(path, params={**p, "label_column": 2, "header": True}), 5)
    in_memory = lgb.train(p, lgb.Dataset(X, label=y, weight=np.full(500, 2.0)), 5)
    assert from_file.model_to_string().split("parameters:")[0] == in_memory.model_to_string().split("parameters:")[0]


def test_estimator_fit_arguments_and_fitted_attributes():
    rng = np.random.RandomState(0)
    X = rng.randn(1000, 4)
    y = X[:, 0] + 0.1 * rng.randn(1000)
    m = LGBMRegressor(n_estimators=20, device_type="cpu").fit(
        X[:800], y[:800], eval_set=[(X[800:], y[800:])], eval_sample_weight=[np.ones(200)],
        eval_metric=lambda yt, yp: ("mad", float(np.median(np.abs(yt - yp))), False),
        feature_name=["a", "b", "c", "d"], early_stopping_rounds=5)
    assert m.feature_name_ == ["a", "b", "c", "d"] and m.n_features_ == 4 and m.objective_ == "regression"
    assert list(m.evals_result_["valid_0"]) == ["l2", "mad"] and m.n_iter_ == 20
    assert set(m.best_score_["valid_0"]) == {"l2", "mad"}


def test_extra_keyword_parameters_survive_clone_and_set_params():
    from sklearn.base import clone
    est = LGBMRegressor(n_estimators=5, device_type="cpu", min_data_per_group=7, path_smooth=2.0)
    assert est.get_params()["min_data_per_group"] == 7
    twin = clone(est)
    assert twin.get_params()["path_smooth"] == 2.0 and twin._params()["min_data_per_group"] == 7
    twin.set_params(path_smooth=0.0, num_leaves=9, cat_l2=1.0)
    assert twin.num_leaves == 9 and twin._params()["path_smooth"] == 0.0 and twin._params()["cat_l2"] == 1.0


def test_cli_weight_group_and_ignored_columns(tmp_path):
    """weight_column / group_column / ignore_column of a CSV file against the same data passed as arrays."""
    from lightgbm_amd import cli
    rng = np.random.RandomState(4)
    n_q, per_q = 60, 12
    n = n_q * per_q
    X = rng.randn(n, 4)
    rel = np.clip((X[:, 0] + 0.5 * X[:, 2] + 0.3 * rng.randn(n)).round() + 1, 0, 3)
    w = rng.uniform(0.5, 2.0, size=n)
    qid = np.repeat(np.arange(n_q), per_q)
    junk = rng.randn(n)
    # label | x0 | weight | x1 | qid | junk | x2 | x3   -> besides the label: weight = 1, qid = 3, junk = 4
    table = np.column_stack([rel, X[:, 0], w, X[:, 1], qid, junk, X[:, 2], X[:, 3]])
    path = tmp_path / "rank.csv"
    np.savetxt(path, table, delimiter=",", fmt="%.10g")
    table = np.loadtxt(path, delimiter=",")            # the values as the file holds them
    X, w = table[:, [1, 3, 6, 7]], table[:, 2]
    out = tmp_path / "m.txt"
    args = ["task=train", f"data={path}", "objective=lambdarank", "num_leaves=7", "min_data_in_leaf=5", "num_iterations=5",
            "weight_column=1", "group_column=3", "ignore_column=4", f"output_model={out}", "device_type=cpu", "verbose=-1"]
    assert cli.main(args) == 0
    want = lgb.train({"objective": "lambdarank", "num_leaves": 7, "min_data_in_leaf": 5, "device_type": "cpu", "verbose": -1},
                     lgb.Dataset(X, label=rel, weight=w, group=np.full(n_q, per_q)), 5)
    got = lgb.Booster(model_file=str(out))
    assert got.num_feature() == 4
    cut = lambda s: s[s.index("Tree=0"):s.index("end of trees")]
    assert cut(got.model_to_string()) == cut(want.model_to_string())
    res = tmp_path / "p.txt"
    assert cli.main(["task=predict", f"data={path}", f"input_model={out}", f"output_result={res}",
                     "weight_column=1", "group_column=3", "ignore_column=4"]) == 0
    np.testing.assert_allclose(np.loadtxt(res), want.predict(X), atol=1e-12)
