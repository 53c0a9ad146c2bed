"""models/: the NumPy model reader as an independent oracle for the native writer, parser and predictor."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.models import TreeEnsemble
from lightgbm_amd.sklearn import LGBMRandomForestRegressor, LGBMRegressor
from lightgbm_amd.utils.datasets import make_regression

CPU = {"device_type": "cpu", "verbose": -1}


def _data(n=4000, f=8, seed=0, nan=True):
    X, y = make_regression(n, f, seed=seed, coef_seed=0)
    if nan:
        X[::11, 1] = np.nan
        X[::7, 3] = np.nan
    return X, y


@pytest.mark.parametrize("objective", ["regression", "binary", "poisson"])
def test_numpy_reader_reproduces_native_predictions(objective):
    X, y = _data()
    if objective == "binary":
        y = (y > np.median(y)).astype(float)
    if objective == "poisson":
        y = np.round(np.exp(0.3 * (y - y.mean()) / y.std())).clip(0)
    b = lgb.train({**CPU, "objective": objective, "num_leaves": 15, "min_data_in_leaf": 10},
                  lgb.Dataset(X, label=y, params=CPU), 25)
    ens = TreeEnsemble.from_booster(b)
    assert ens.num_trees == 25 and ens.objective.split()[0] == objective
    Xt, _ = _data(1500, 8, seed=5)
    np.testing.assert_allclose(ens.predict(Xt), b.predict(Xt), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ens.predict(Xt, raw_score=True, num_iteration=7), b.predict(Xt, raw_score=True, num_iteration=7),
                               rtol=1e-12, atol=1e-12)
    # the learned default direction for NaN is part of the file
    assert any(((t.decision_type >> 2) & 3 == 2).any() for t in ens.trees)


def test_text_round_trip_is_stable_and_counts_are_consistent(tmp_path):
    X, y = _data(3000, 6, seed=2, nan=False)
    b = lgb.train({**CPU, "num_leaves": 31}, lgb.Dataset(X, label=y, params=CPU), 10)
    path = tmp_path / "m.txt"
    b.save_model(str(path))
    again = lgb.Booster(model_file=str(path))
    assert again.model_to_string() == b.model_to_string()          # parse -> write is the identity
    ens = TreeEnsemble.from_file(str(path))
    for t in ens.trees:
        assert t.leaf_count.sum() == 3000 and t.internal_count[0] == 3000
        # every internal node's count is the sum of its children's
        for nd in range(t.num_leaves - 1):
            kids = 0
            for c in (t.left_child[nd], t.right_child[nd]):
                kids += t.internal_count[c] if c >= 0 else t.leaf_count[~c]
            assert kids == t.internal_count[nd]
    leaves = ens.predict(X, pred_leaf=True)
    assert leaves.shape == (3000, 10)
    np.testing.assert_array_equal(np.bincount(leaves[:, 0], minlength=ens.trees[0].num_leaves), ens.trees[0].leaf_count)
    df = ens.to_dataframe()
    assert len(df) == sum(2 * t.num_leaves - 1 for t in ens.trees)
    assert set(df.columns) >= {"tree_index", "node_index", "split_feature", "threshold", "value", "count"}
    assert max(ens.depths()) >= 3
    assert ens.parameters["num_leaves"] == "31"


def test_random_forest_family_averages_and_stages():
    X, y = _data(5000, 8, seed=3, nan=False)
    Xt, yt = _data(2000, 8, seed=4, nan=False)
    rf = LGBMRandomForestRegressor(n_estimators=40, num_leaves=127, min_child_samples=5, colsample_bytree=0.8, device_type="cpu",
                                   bagging_seed=77, feature_fraction_seed=5).fit(X, y)
    gb = LGBMRegressor(n_estimators=40, device_type="cpu", learning_rate=0.2).fit(X, y)
    ens = TreeEnsemble.from_booster(rf.booster_)
    assert ens.average_output
    np.testing.assert_allclose(ens.predict(Xt), rf.predict(Xt), rtol=1e-12, atol=1e-12)
    staged = list(rf.staged_predict(Xt, step=10))
    assert len(staged) == 4
    err = [float(np.sqrt(np.mean((p - yt) ** 2))) for p in staged]
    assert err[-1] <= err[0] * 1.02                       # more bagged trees never hurt much
    np.testing.assert_allclose(staged[-1], rf.predict(Xt))
    base = float(np.std(yt))
    assert float(np.sqrt(np.mean((gb.predict(Xt) - yt) ** 2))) < 0.6 * base
    assert err[-1] < 0.8 * base


def test_tree_sizes_header_gives_the_byte_length_of_every_tree_block():
    """Upstream's loader cuts the tree section by `tree_sizes`; the blocks must have exactly these lengths."""
    X, y = _data(2000, 5, seed=9, nan=False)
    b = lgb.train({**CPU, "num_leaves": 7}, lgb.Dataset(X, label=y, params=CPU), 5)
    text = b.model_to_string()
    sizes = [int(v) for v in text.split("tree_sizes=", 1)[1].splitlines()[0].split()]
    body = text[text.index("Tree=0"):text.index("end of trees")]
    assert sum(sizes) == len(body.encode())
    pos = 0
    for i, sz in enumerate(sizes):
        block = body[pos:pos + sz]
        assert block.startswith(f"Tree={i}\n") and block.endswith("\n\n")
        pos += sz


# ----------------------------------------------------------------------------- interpretation / R helpers
def test_shap_contributions_equal_brute_force_shapley():
    """pred_contrib against the definition: Shapley values of the path-dependent conditional expectation
    (unknown features average the children by their training row counts), all 2^F subsets enumerated."""
    import itertools, math
    X, y = make_regression(3000, 4, seed=1)
    X[::11, 2] = np.nan
    b = lgb.train({**CPU, "num_leaves": 12, "min_data_in_leaf": 5}, lgb.Dataset(X, label=y, params=CPU), 6)
    phi = b.predict(X[:12], pred_contrib=True)
    np.testing.assert_allclose(phi.sum(axis=1), b.predict(X[:12], raw_score=True), atol=1e-12)
    trees = b.dump_model()["tree_info"]
    F = 4

    def count(n):
        return n["internal_count"] if "split_index" in n else n["leaf_count"]

    def expect(n, x, known):
        if "split_index" not in n:
            return n["leaf_value"]
        f = n["split_feature"]
        if f in known:
            v = x[f]
            left = n["default_left"] if np.isnan(v) else v <= n["threshold"]
            return expect(n["left_child"] if left else n["right_child"], x, known)
        l, r = n["left_child"], n["right_child"]
        return (count(l) * expect(l, x, known) + count(r) * expect(r, x, known)) / (count(l) + count(r))

    for i in range(6):
        want = np.zeros(F + 1)
        for t in trees:
            root = t["tree_structure"]
            want[F] += expect(root, X[i], set())
            for f in range(F):
                others = [j for j in range(F) if j != f]
                for k in range(F):
                    w = math.factorial(k) * math.factorial(F - k - 1) / math.factorial(F)
                    for S in itertools.combinations(others, k):
                        want[f] += w * (expect(root, X[i], set(S) | {f}) - expect(root, X[i], set(S)))
        np.testing.assert_allclose(phi[i], want, atol=1e-12)


def test_r_style_helper_surface(tmp_path):
    import pickle
    from lightgbm_amd import rstyle as R
    X, y = make_regression(2000, 5, seed=2)
    d = R.lgb_Dataset(X[:1500], label=y[:1500], params=CPU)
    v = R.lgb_Dataset_create_valid(d, X[1500:], label=y[1500:])
    assert R.dim(d) == (1500, 5) and R.dim(R.slice_rows(d, range(1, 101))) == (100, 5)
    np.testing.assert_allclose(R.getinfo(d, "label"), y[:1500].astype(np.float32))
    fit = R.lgb_train({"learning_rate": 0.1, **CPU}, d, 15, valids={"test": v}, objective="regression", eval="rmse",
                      verbose=0)
    curve = R.lgb_get_eval_result(fit, "test", "rmse")
    assert len(curve) == 15 and curve[-1] < curve[0]
    assert R.lgb_get_eval_result(fit, "test", "rmse", iters=[1, 15]).tolist() == [curve[0], curve[-1]]
    imp = R.lgb_importance(fit)
    assert list(imp.columns) == ["Feature", "Gain", "Cover", "Frequency"] and abs(imp["Gain"].sum() - 1) < 1e-12
    assert imp["Gain"].is_monotonic_decreasing
    tab = R.lgb_model_dt_tree(fit)
    assert {"tree_index", "depth", "node_parent", "split_feature", "threshold", "value", "count"} <= set(tab.columns)
    assert tab["tree_index"].nunique() == 15
    assert (tab[tab["split_feature"].isna()].groupby("tree_index")["count"].sum() == 1500).all()
    inter = R.lgb_interprete(fit, X, [1, 2, 3])
    assert len(inter) == 3 and list(inter[0].columns) == ["Feature", "Contribution"]
    assert inter[0]["Contribution"].abs().is_monotonic_decreasing
    # model files, JSON dump, object serialisation
    R.lgb_save(fit, str(tmp_path / "m.txt"))
    again = R.lgb_load(str(tmp_path / "m.txt"))
    np.testing.assert_array_equal(R.predict(again, X[1500:]), R.predict(fit, X[1500:]))
    assert '"tree_info"' in R.lgb_dump(fit)
    R.saveRDS_lgb_Booster(fit, str(tmp_path / "m.rds"))
    np.testing.assert_array_equal(R.predict(R.readRDS_lgb_Booster(str(tmp_path / "m.rds")), X[:50]), R.predict(fit, X[:50]))
    clone = pickle.loads(pickle.dumps(fit))
    assert clone.model_to_string() == fit.model_to_string()
    assert fit.lower_bound() <= R.predict(fit, X).min() and R.predict(fit, X).max() <= fit.upper_bound()
    leaf = R.predict(fit, X[:5], predleaf=True)
    assert leaf.shape == (5, 15) and fit.get_leaf_output(0, int(leaf[0, 0])) == fit._tree_arrays()[0]["leaf_value"][leaf[0, 0]]


def test_refit_keeps_the_structure_and_moves_the_leaf_values():
    rng = np.random.RandomState(0)
    X = rng.randn(6000, 6)
    y = X[:, 0] + np.sin(X[:, 1]) + 0.1 * rng.randn(6000)
    b = lgb.train({"device_type": "cpu", "verbose": -1, "num_leaves": 15}, lgb.Dataset(X[:3000], label=y[:3000]), 30)
    l2 = lambda m, A, t: float(np.mean((m.predict(A) - t) ** 2))
    # on its own training data the Newton steps are (up to the histogram quantisation) the ones it was built with
    same = b.refit(X[:3000], y[:3000], decay_rate=0.0)
    assert np.abs(same.predict(X) - b.predict(X)).max() < 0.01
    # a shifted target is picked up without touching a single split
    r = b.refit(X[3000:], y[3000:] + 2.0, decay_rate=0.0)
    assert l2(r, X[3000:], y[3000:] + 2.0) < 0.05 < l2(b, X[3000:], y[3000:] + 2.0)
    structure = lambda m: m.trees_to_dataframe()[["tree_index", "node_index", "split_feature", "threshold"]].fillna(-1).to_dict("list")
    assert structure(r) == structure(b)
    half = b.refit(X[3000:], y[3000:] + 2.0, decay_rate=0.5)
    assert l2(r, X[3000:], y[3000:] + 2.0) < l2(half, X[3000:], y[3000:] + 2.0) < l2(b, X[3000:], y[3000:] + 2.0)
    np.testing.assert_array_equal(b.refit(X[3000:], y[3000:], decay_rate=1.0).predict(X), b.predict(X))
    # multiclass and binary go through their own gradients
    yc = (X[:, 0] > 0).astype(float) + (X[:, 1] > 0.5)
    mc = lgb.train({"device_type": "cpu", "verbose": -1, "objective": "multiclass", "num_class": 3}, lgb.Dataset(X[:3000], label=yc[:3000]), 10)
    rc = mc.refit(X[3000:], yc[3000:], decay_rate=0.3)
    assert rc.predict(X).shape == (6000, 3) and (rc.predict(X[3000:]).argmax(axis=1) == yc[3000:]).mean() > 0.9
    with pytest.raises(lgb.LightGBMError):
        b.refit(X[:, :3], y)


def test_booster_and_dataset_housekeeping_methods():
    rng = np.random.RandomState(0)
    X = rng.randn(2000, 4)
    y = X[:, 0] + 0.1 * rng.randn(2000)
    d = lgb.Dataset(X, label=y, free_raw_data=False)
    b = lgb.train({"device_type": "cpu", "verbose": -1, "num_leaves": 7}, d, 10)
    everything = b.predict(X)
    staged = b.predict(X, num_iteration=3)
    b.shuffle_models(seed=3)
    np.testing.assert_allclose(b.predict(X), everything, atol=1e-12)       # a sum does not care about the order
    assert not np.allclose(b.predict(X, num_iteration=3), staged)
    old = b.get_leaf_output(2, 1)
    b.set_leaf_output(2, 1, old + 1.0)
    assert b.get_leaf_output(2, 1) == pytest.approx(old + 1.0)
    with pytest.raises(lgb.LightGBMError):
        b.set_leaf_output(99, 0, 0.0)
    b.set_attr(note="x")
    assert b.attr("note") == "x" and b.set_attr(note=None).attr("note") is None
    with pytest.raises(lgb.LightGBMError, match="comm="):
        b.set_network("a,b")
    # dataset side
    d.add_features_from(lgb.Dataset(rng.randn(2000, 2), free_raw_data=False))
    assert d.construct().num_feature() == 6 and d.get_feature_name()[4:] == ["D2_Column_0", "D2_Column_1"]
    assert d.feature_num_bin(5) > 10 and d.get_field("label").shape == (2000,)
    v = lgb.Dataset(X[:10]).set_reference(d)
    assert v.get_ref_chain() == {v, d}
    with pytest.raises(lgb.LightGBMError):
        d.set_reference(v)
    d.set_field("weight", np.ones(2000))
    assert d.get_weight().sum() == 2000
