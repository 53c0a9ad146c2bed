"""Categorical features: category codes get one bin each (by frequency).  Up to max_cat_to_onehot categories
a split sends ONE category left (one against the rest); above it the categories are sorted by their gradient
statistics and a split sends a SET of them left (cat_smooth / cat_l2 / max_cat_threshold /
min_data_per_group).  The model file uses upstream's cat_boundaries / cat_threshold bitsets."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.models import TreeEnsemble

CPU = {"device_type": "cpu", "verbose": -1}


def make_data(n=6000, seed=0):
    rng = np.random.RandomState(seed)
    city = rng.randint(0, 12, size=n)                      # 12 categories, effect unrelated to the code's order
    effect = np.array([3.0, -2.0, 0.5, 4.0, -3.5, 0.0, 1.5, -1.0, 2.5, -4.0, 0.8, -0.3])
    x1 = rng.randn(n)
    y = effect[city] + 0.7 * x1 + 0.2 * rng.randn(n)
    X = np.c_[x1, city.astype(float), rng.randn(n)]
    return X, y, city, effect


def test_category_set_splits_beat_one_against_the_rest_on_many_categories():
    rng = np.random.RandomState(0)
    n = 20000
    codes = rng.randint(0, 60, size=n)
    effect = rng.randn(60) * 2
    X = np.column_stack([codes, rng.randn(n, 3)])
    y = effect[codes] + 0.5 * X[:, 1] + 0.3 * rng.randn(n)
    base = {**CPU, "num_leaves": 15, "min_data_in_leaf": 20}
    l2 = lambda m: float(np.mean((m.predict(X) - y) ** 2))
    sets = lgb.train(base, lgb.Dataset(X, label=y, categorical_feature=[0]), 20)
    single = lgb.train({**base, "max_cat_to_onehot": 256}, lgb.Dataset(X, label=y, categorical_feature=[0]), 20)
    assert l2(sets) < 0.5 * l2(single)
    size = lambda m: [len(str(t).split("||")) for t in m.trees_to_dataframe().query("decision_type == '=='")["threshold"]]
    assert max(size(single)) == 1
    assert max(size(sets)) > 5 and max(size(sets)) <= 30              # (60 + 1) / 2 caps a set at 30 categories
    assert max(size(lgb.train({**base, "max_cat_threshold": 4}, lgb.Dataset(X, label=y, categorical_feature=[0]), 5))) <= 4
    # the sets survive the model file, the NumPy reader and the binned traversal of the training scores
    text = sets.model_to_string()
    np.testing.assert_array_equal(lgb.Booster(model_str=text).predict(X), sets.predict(X))
    np.testing.assert_allclose(TreeEnsemble.from_string(text).predict(X), sets.predict(X), atol=1e-12)
    kept = lgb.Booster(base, lgb.Dataset(X, label=y, categorical_feature=[0]))
    ev = lgb.Dataset(X[:3000], label=y[:3000], reference=kept.train_set)
    kept.add_valid(ev, "v")
    for _ in range(6):
        kept.update()
    np.testing.assert_allclose(kept._inner_predict(-1, raw=True), kept.predict(X, raw_score=True), atol=1e-9)
    np.testing.assert_allclose(kept._inner_predict(0, raw=True), kept.predict(X[:3000], raw_score=True), atol=1e-9)
    kept.rollback_one_iter()
    np.testing.assert_allclose(kept._inner_predict(-1, raw=True), kept.predict(X, raw_score=True), atol=1e-9)
    # a tiny category never decides a split on its own: with huge min_data_per_group no set qualifies
    none = lgb.train({**base, "min_data_per_group": 15000}, lgb.Dataset(X, label=y, categorical_feature=[0]), 3)
    assert "==" not in set(none.trees_to_dataframe()["decision_type"].dropna())


def test_categorical_splits_isolate_categories_and_beat_the_ordinal_reading():
    X, y, city, effect = make_data()
    p = {**CPU, "num_leaves": 31, "min_data_in_leaf": 5, "learning_rate": 0.2, "max_cat_to_onehot": 12}
    cat = lgb.train(p, lgb.Dataset(X, label=y, categorical_feature=[1], params=CPU), 30)
    num = lgb.train(p, lgb.Dataset(X, label=y, params=CPU), 30)
    rm = lambda b: float(np.sqrt(np.mean((b.predict(X) - y) ** 2)))
    assert rm(cat) < 0.35 and rm(cat) < rm(num) + 0.05
    df = cat.trees_to_dataframe()
    cat_nodes = df[(df["split_feature"] == "Column_1")]
    assert len(cat_nodes) > 10 and (cat_nodes["decision_type"] == "==").all()
    assert all("||" not in str(t) and 0 <= int(t) < 12 for t in cat_nodes["threshold"])
    text = cat.model_to_string()
    assert "cat_boundaries=" in text and "cat_threshold=" in text and "num_cat=0" not in text.split("Tree=1")[0].split("Tree=0")[1]
    # feature_infos lists the categories by frequency
    infos = [ln for ln in text.splitlines() if ln.startswith("feature_infos=")][0].split("=")[1].split()
    assert sorted(int(c) for c in infos[1].split(":")) == list(range(12))
    # unseen / negative / missing categories take the right branch everywhere: same prediction for all three
    probe = np.tile(X[:1], (3, 1))
    probe[:, 1] = [99.0, -1.0, np.nan]
    out = cat.predict(probe)
    assert out[0] == out[1] == out[2]
    # round trip, NumPy reader, binned training scores == model predictions, contributions add up
    again = lgb.Booster(model_str=text)
    np.testing.assert_array_equal(again.predict(X), cat.predict(X))
    assert again.model_to_string() == text
    np.testing.assert_allclose(TreeEnsemble.from_string(text).predict(X), cat.predict(X), atol=1e-12)
    kept = lgb.train(p, lgb.Dataset(X, label=y, categorical_feature=[1], params=CPU), 5, keep_training_booster=True)
    np.testing.assert_allclose(kept._inner_predict(-1, raw=True), kept.predict(X, raw_score=True), atol=1e-9)
    phi = cat.predict(X[:20], pred_contrib=True)
    np.testing.assert_allclose(phi.sum(axis=1), cat.predict(X[:20], raw_score=True), atol=1e-9)


def test_validation_sets_cv_and_pandas_categories():
    import pandas as pd
    X, y, city, _ = make_data(4000, seed=1)
    names = np.array([f"c{i:02d}" for i in range(12)])
    df = pd.DataFrame({"x1": X[:, 0], "city": pd.Categorical(names[city]), "noise": X[:, 2]})
    d = lgb.Dataset(df[:3000], label=y[:3000], params=CPU)           # "auto": the category column is found
    v = d.create_valid(df[3000:], label=y[3000:])
    ev = {}
    b = lgb.train({**CPU, "num_leaves": 15, "min_data_in_leaf": 5, "metric": "rmse"}, d, 40, valid_sets=[v], evals_result=ev)
    assert ev["valid_0"]["rmse"][-1] < 0.45
    # prediction frames are coded with the training levels, also after a save / load and with shuffled level order
    p1 = b.predict(df[3000:])
    assert abs(np.sqrt(np.mean((p1 - y[3000:]) ** 2)) - ev["valid_0"]["rmse"][-1]) < 1e-6
    shuffled = df[3000:].copy()
    shuffled["city"] = pd.Categorical(shuffled["city"].astype(str), categories=list(names[::-1]))
    np.testing.assert_array_equal(b.predict(shuffled), p1)
    np.testing.assert_array_equal(lgb.Booster(model_str=b.model_to_string()).predict(shuffled), p1)
    r = lgb.cv({**CPU, "num_leaves": 15, "min_data_in_leaf": 5, "metric": "rmse"}, lgb.Dataset(df, label=y, params=CPU), 20, nfold=3)
    assert r["rmse-mean"][-1] < 0.6
    by_name = lgb.train({**CPU, "num_leaves": 15}, lgb.Dataset(df.assign(city=city.astype(float)), label=y,
                                                                 categorical_feature=["city"], params=CPU), 5)
    assert "==" in set(by_name.trees_to_dataframe()["decision_type"].dropna())


def test_upstream_style_bitset_with_several_categories_parses_and_predicts():
    """A node whose set holds categories {1, 4, 35} (two bitset words), as upstream's many-vs-many search writes."""
    text = "\n".join([
        "tree", "version=v4", "num_class=1", "num_tree_per_iteration=1", "label_index=0", "max_feature_idx=1",
        "objective=regression", "feature_names=a b", "feature_infos=[0:1] 1:4:35:2", "tree_sizes=300", "",
        "Tree=0", "num_leaves=2", "num_cat=1", "split_feature=1", "split_gain=1", "threshold=0", "decision_type=1",
        "left_child=-1", "right_child=-2", "leaf_value=10 20", "leaf_weight=1 1", "leaf_count=1 1",
        "internal_value=0", "internal_weight=2", "internal_count=2", "cat_boundaries=0 2",
        f"cat_threshold={(1 << 1) | (1 << 4)} {1 << 3}", "is_linear=0", "shrinkage=1", "", "",
        "end of trees", "", "feature_importances:", "b=1", "", "parameters:", "end of parameters", "",
        "pandas_categorical:null", ""])
    b = lgb.Booster(model_str=text)
    X = np.array([[0, 1], [0, 4], [0, 35], [0, 2], [0, 0], [0, 36], [0, 64], [0, -3], [0, np.nan]], dtype=float)
    want = [10, 10, 10, 20, 20, 20, 20, 20, 20]
    assert b.predict(X).tolist() == want
    assert TreeEnsemble.from_string(text).predict(X).tolist() == want
    assert b.dump_model()["tree_info"][0]["tree_structure"]["threshold"] == "1||4||35"
    assert lgb.Booster(model_str=b.model_to_string()).predict(X).tolist() == want
