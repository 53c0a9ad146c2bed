"""Bin construction and value->bin mapping."""
import numpy as np
import pytest

import lightgbm_amd as lgb

CPU = {"device_type": "cpu", "verbose": -1}


def _ds(X, **params):
    d = lgb.Dataset(X, label=np.zeros(len(X)), params={"device_type": "cpu", **params})
    d.construct()
    return d


def test_distinct_values_get_one_bin_each():
    rng = np.random.RandomState(0)
    col = rng.choice([1.0, 2.0, 3.0, 4.0, 5.0], size=1000)     # an ordinal like `cut`
    d = _ds(np.column_stack([col, rng.randn(1000)]), min_data_in_bin=1)
    info = d.bin_info(0)
    assert info["num_bin"] == 5 and info["missing_bin"] == -1
    np.testing.assert_allclose(info["upper_bounds"][:4], [1.5, 2.5, 3.5, 4.5])
    assert np.isinf(info["upper_bounds"][-1])
    bins = d.get_bins()
    np.testing.assert_array_equal(bins[0], (col - 1).astype(np.uint8))


def test_equal_frequency_bins_and_max_bin():
    rng = np.random.RandomState(1)
    x = rng.randn(20000)
    d = _ds(x.reshape(-1, 1), max_bin=63)
    info = d.bin_info(0)
    assert 50 <= info["num_bin"] <= 63
    counts = np.bincount(d.get_bins()[0], minlength=info["num_bin"])
    assert counts.min() > 0 and counts.max() < 3 * counts.mean()
    ub = info["upper_bounds"]
    assert np.all(np.diff(ub[:-1]) > 0)
    # first i with x <= upper[i]
    expect = np.searchsorted(ub, x, side="left")
    np.testing.assert_array_equal(d.get_bins()[0], expect.astype(np.uint8))


def test_nan_gets_its_own_last_bin_and_constant_feature_is_dropped():
    rng = np.random.RandomState(2)
    x = rng.randn(3000)
    x[::7] = np.nan
    X = np.column_stack([x, np.full(3000, 3.0), rng.randn(3000)])
    d = _ds(X)
    info = d.bin_info(0)
    assert info["missing_bin"] == info["num_bin"] - 1
    bins = d.get_bins()
    assert bins.shape[0] == 2                     # the constant column is not a used feature
    assert np.all(bins[0][::7] == info["missing_bin"])
    assert d.bin_info(1)["num_bin"] == 1


def test_reference_shares_boundaries_and_subset_shares_bins():
    rng = np.random.RandomState(3)
    X = rng.randn(5000, 4)
    y = X[:, 0]
    d = lgb.Dataset(X, label=y, params={"device_type": "cpu"})
    v = d.create_valid(X[:1000] * 2.0, label=y[:1000])
    v.construct()
    for f in range(4):
        np.testing.assert_array_equal(d.bin_info(f)["upper_bounds"], v.bin_info(f)["upper_bounds"])
    idx = np.arange(0, 5000, 3)
    s = d.subset(idx)
    assert s.num_data() == len(idx) and s.num_feature() == 4
    np.testing.assert_allclose(s.get_label(), y[idx], rtol=1e-6)


def test_bin_sample_controls_boundaries():
    rng = np.random.RandomState(4)
    X = rng.randn(4000, 3)
    a = lgb.Dataset(X[:2000], label=np.zeros(2000), bin_sample=X, params={"device_type": "cpu"}).construct()
    b = lgb.Dataset(X[2000:], label=np.zeros(2000), bin_sample=X, params={"device_type": "cpu"}).construct()
    for f in range(3):
        np.testing.assert_array_equal(a.bin_info(f)["upper_bounds"], b.bin_info(f)["upper_bounds"])


def test_binary_cache_round_trip(tmp_path):
    """save_binary / Dataset(path): same bins, same fields, same model - without the raw matrix."""
    rng = np.random.RandomState(3)
    X = rng.randn(3000, 40)
    X[::6, 2] = np.nan
    X[:, 9] = 7.0
    y = X[:, 0] - 2 * X[:, 1] + rng.randn(3000) * 0.1
    w = rng.rand(3000) + 0.5
    d = lgb.Dataset(X, label=y, weight=w, params=CPU, feature_name=[f"f{i}" for i in range(40)])
    path = tmp_path / "train.bin"
    d.save_binary(str(path))
    e = lgb.Dataset(str(path), params=CPU).construct()
    assert e.num_data() == 3000 and e.num_feature() == 40
    np.testing.assert_array_equal(e.get_bins(), d.get_bins())
    np.testing.assert_array_equal(e.get_label(), d.get_label())
    np.testing.assert_array_equal(e.get_weight(), d.get_weight())
    assert e.get_feature_name() == d.get_feature_name()
    for f in (0, 2, 9):
        a, b = d.bin_info(f), e.bin_info(f)
        assert a["num_bin"] == b["num_bin"] and a["missing_bin"] == b["missing_bin"]
        np.testing.assert_array_equal(a["upper_bounds"], b["upper_bounds"])
    p = {**CPU, "num_leaves": 15}
    assert lgb.train(p, d, 5).model_to_string() == lgb.train(p, e, 5).model_to_string()
    with open(path, "r+b") as fh:      # a damaged file is refused, not half-loaded
        fh.truncate(1000)
    with pytest.raises(lgb.LightGBMError):
        lgb.Dataset(str(path), params=CPU).construct()


def test_max_bin_by_feature():
    import lightgbm_amd as lgb
    rng = np.random.RandomState(0)
    X = rng.randn(5000, 3)
    d = lgb.Dataset(X, label=X[:, 0], params={"device_type": "cpu", "verbose": -1, "max_bin_by_feature": [4, 255, 16]}).construct()
    assert [d.feature_num_bin(f) for f in range(3)] == [4, 255, 16]
    import pytest
    with pytest.raises(lgb.LightGBMError, match="max_bin_by_feature"):
        lgb.Dataset(X, label=X[:, 0], params={"device_type": "cpu", "max_bin_by_feature": [4, 255]}).construct()
