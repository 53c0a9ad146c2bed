"""models/: the NumPy model reader as an independent oracle for the native writer, parser and predictor."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.models import GBDT, RandomForest, TreeEnsemble
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
    rf = RandomForest(n_estimators=40, device_type="cpu", seed=1).fit(X, y)
    gb = GBDT(n_estimators=40, device_type="cpu", learning_rate=0.2).fit(X, y)
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
