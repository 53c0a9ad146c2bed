"""MI355X tests: every hot kernel against a NumPy/PyTorch reference, and the device engine against the CPU oracle.

Run with ``pytest -m gpu`` on a GPU box.  The HIP core must be the loaded library (no silent fallback).
"""
import threading

import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like, make_higgs_like, make_regression

pytestmark = pytest.mark.gpu

GPU = {"device_type": "gpu", "verbose": -1}
CPU = {"device_type": "cpu", "verbose": -1}


def trees_of(b):
    s = b.model_to_string() if hasattr(b, "model_to_string") else b
    return s[s.index("Tree=0"):s.index("end of trees")]


def test_hip_core_is_loaded():
    from lightgbm_amd import _lib
    lib, hip = _lib.load()
    assert hip and lib.lga_build_info() == b"hip"


# ----------------------------------------------------------------------------- kernels
def _random_bins(n, F, seed, low_card=()):
    rng = np.random.RandomState(seed)
    T = (F + 31) // 32
    bins = rng.randint(0, 256, size=(n, F)).astype(np.uint8)
    for f in low_card:
        bins[:, f] = rng.randint(0, 5, size=n)
    rt = np.zeros((T, n, 32), dtype=np.uint8)
    for f in range(F):
        rt[f // 32, :, f % 32] = bins[:, f]
    return bins, rt


@pytest.mark.parametrize("n,F,chunk", [(1000, 28, 4096), (70000, 28, 4096), (50000, 70, 16384), (3, 5, 1024)])
def test_histogram_kernel_matches_numpy(n, F, chunk):
    import torch
    from lightgbm_amd import ops
    N = n + 500
    bins, rt = _random_bins(N, F, seed=n, low_card=(1, 3))
    rng = np.random.RandomState(1)
    g = rng.randn(N).astype(np.float32)
    h = rng.rand(N).astype(np.float32)
    packed, kg, kh = ops.pack_gradients(g, h)
    rows = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
    gq, hq = packed >> 32, packed & 0xffffffff
    G, H, hd = ops.hist_build(torch.from_numpy(rt).cuda(), torch.from_numpy(rows).cuda(),
                              torch.from_numpy(packed).cuda(), F, rows_per_chunk=chunk)
    G, H, hd = G.cpu().numpy(), H.cpu().numpy(), hd.cpu().numpy()
    for f in range(F):
        eg = np.bincount(bins[rows, f], weights=None, minlength=256) * 0
        eg = np.zeros(256, dtype=np.int64); eh = np.zeros(256, dtype=np.int64)
        np.add.at(eg, bins[rows, f], gq[rows]); np.add.at(eh, bins[rows, f], hq[rows])
        np.testing.assert_array_equal(G[f], eg)      # integer sums: exact
        np.testing.assert_array_equal(H[f], eh)
    assert G[F:].sum() == 0 and H[F:].sum() == 0     # padding features untouched
    assert hd[0] == n and hd[1] == gq[rows].sum() and hd[2] == hq[rows].sum()


@pytest.mark.parametrize("n,F,unit,fused_quant", [(1000, 28, 2048, False), (70000, 28, 4096, True), (50000, 6, 16384, False),
                                                  (3, 5, 2048, True), (40000, 32, 2048, True)])
def test_row_line_histogram_kernel_matches_numpy(n, F, unit, fused_quant):
    """Root pass of the fused growth kernel (row lines, rotated records, optional in-pass quantisation)."""
    import torch
    from lightgbm_amd import ops
    N = n + 500
    bins, rt = _random_bins(N, F, seed=n + 1, low_card=(1, 3))
    rng = np.random.RandomState(2)
    g = rng.randn(N).astype(np.float32)
    h = rng.rand(N).astype(np.float32)
    rows = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
    if fused_quant:
        G, H, hd, scales = ops.lines_hist_build(torch.from_numpy(rt).cuda(), torch.from_numpy(rows).cuda(), F,
                                                g=torch.from_numpy(g).cuda(), h=torch.from_numpy(h).cuda(), unit_rows=unit)
        kg, kh = ops.quant_exponent(float(np.abs(g[rows]).max())), ops.quant_exponent(float(h[rows].max()))
        assert scales == (kg, kh)
        lim = float(1 << ops.QUANT_BITS)
        gq = np.rint(np.clip(np.ldexp(g, kg), -lim, lim)).astype(np.int64)
        hq = np.rint(np.clip(np.ldexp(h, kh), -lim, lim)).astype(np.int64)
    else:
        packed, kg, kh = ops.pack_gradients(g, h)
        gq, hq = packed >> 32, packed & 0xffffffff
        G, H, hd, _ = ops.lines_hist_build(torch.from_numpy(rt).cuda(), torch.from_numpy(rows).cuda(), F,
                                           packed=torch.from_numpy(packed).cuda(), unit_rows=unit)
    G, H, hd = G.cpu().numpy(), H.cpu().numpy(), hd.cpu().numpy()
    for f in range(F):
        eg = np.zeros(256, dtype=np.int64); eh = np.zeros(256, dtype=np.int64)
        np.add.at(eg, bins[rows, f], gq[rows]); np.add.at(eh, bins[rows, f], hq[rows])
        np.testing.assert_array_equal(G[f], eg)      # integer sums: exact
        np.testing.assert_array_equal(H[f], eh)
    assert G[F:].sum() == 0 and H[F:].sum() == 0     # padding features untouched
    assert hd[0] == n and hd[1] == gq[rows].sum() and hd[2] == hq[rows].sum()


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100000])
def test_partition_kernel_is_a_stable_split(n):
    import torch
    from lightgbm_amd import ops
    rng = np.random.RandomState(n)
    N = n + 300
    fm = rng.randint(0, 40, size=(3, N)).astype(np.uint8)
    rows = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
    for (f, thr, mb, dl) in [(1, 17, -1, False), (2, 9, 39, True), (0, 0, 39, False)]:
        out, nl = ops.partition_rows(torch.from_numpy(rows).cuda(), torch.from_numpy(fm).cuda(), f, thr, mb, dl)
        b = fm[f, rows]
        left = np.where(b == mb, dl, b <= thr)
        np.testing.assert_array_equal(out.cpu().numpy(), np.concatenate([rows[left], rows[~left]]))   # stable: exact
        assert nl == int(left.sum())


@pytest.mark.parametrize("n,block_rows", [(1, 2048), (4097, 2048), (100000, 2048), (100000, 4096), (300001, 8192)])
def test_single_pass_partition_kernel_splits_exactly(n, block_rows):
    """The growth rounds' single-pass partition (rows in registers, one reservation per workgroup, lefts forward and
    rights backward): the left part holds exactly the rows that go left, the right part the others - as sets, the
    order inside a child carries no meaning - and the children's ranges and the left count are what the split says."""
    import torch
    from lightgbm_amd import ops
    rng = np.random.RandomState(n)
    N = n + 500
    fm = rng.randint(0, 40, size=(3, N)).astype(np.uint8)
    rows = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
    for (f, thr, mb, dl) in [(1, 17, -1, False), (2, 9, 39, True), (0, 0, 39, False), (1, 39, -1, False)]:
        out, nl = ops.partition_rows_single_pass(torch.from_numpy(rows).cuda(), torch.from_numpy(fm).cuda(), f, thr, mb, dl, block_rows)
        b = fm[f, rows]
        left = np.where(b == mb, dl, b <= thr)
        out = out.cpu().numpy()
        assert nl == int(left.sum())
        np.testing.assert_array_equal(np.sort(out[:nl]), rows[left])
        np.testing.assert_array_equal(np.sort(out[nl:]), rows[~left])


def _scan_reference(G, H, cnt, nb, mb, kg, kh, l1, l2, min_gain, min_h, min_data):
    gu, hu = 2.0 ** -kg, 2.0 ** -kh
    tg, th = int(G[0, :].sum()), int(H[0, :].sum())
    factor = cnt / th
    def lg(g, hh):
        t = np.sign(g) * max(abs(g) - l1, 0.0)
        return t * t / (hh + l2)
    parent = lg(tg * gu, th * hu)
    best = (-np.inf, None)
    for f in range(len(nb)):
        vb = nb[f] - 1 if mb[f] >= 0 else nb[f]
        cg, ch = np.cumsum(G[f, :vb]), np.cumsum(H[f, :vb])
        last = vb - 1 if mb[f] >= 0 else vb - 2
        for t in range(last + 1):
            cands = [(int(cg[t]), int(ch[t]), 0)]
            if mb[f] >= 0 and t <= vb - 2:
                cands.append((int(cg[t] + G[f, mb[f]]), int(ch[t] + H[f, mb[f]]), 1))
            for a, b, dl in cands:
                lc = int(b * factor + 0.5); rc = cnt - lc
                if lc < min_data or rc < min_data or b * hu < min_h or (th - b) * hu < min_h:
                    continue
                gain = lg(a * gu, b * hu) + lg((tg - a) * gu, (th - b) * hu) - parent
                if not gain > min_gain:
                    continue
                key = (f << 16) | (t << 1) | dl
                if gain > best[0] or (gain == best[0] and key < best[1][0]):
                    best = (gain, (key, f, t, dl, a, b))
    return best


@pytest.mark.parametrize("seed,l1,l2", [(0, 0.0, 0.0), (1, 0.5, 1.0), (2, 0.0, 10.0)])
def test_scan_kernel_picks_the_reference_split(seed, l1, l2):
    import torch
    from lightgbm_amd import ops
    rng = np.random.RandomState(seed)
    F = 9
    nb = np.array([256, 5, 64, 256, 17, 2, 100, 255, 33])
    mb = np.array([-1, -1, 63, 255, -1, -1, -1, -1, 32])
    n = 20000
    g = rng.randn(n).astype(np.float32); h = (rng.rand(n) + 0.1).astype(np.float32)
    packed, kg, kh = ops.pack_gradients(g, h)
    gq, hq = packed >> 32, packed & 0xffffffff
    G = np.zeros((32, 256), dtype=np.int64); H = np.zeros((32, 256), dtype=np.int64)
    for f in range(F):
        b = rng.randint(0, nb[f], size=n)
        if f == 0:
            gq2 = gq + (b < 100) * 3000        # plant a strong split
            np.add.at(G[f], b, gq2); np.add.at(H[f], b, hq)
            G[f] -= 0
        else:
            np.add.at(G[f], b, gq); np.add.at(H[f], b, hq)
    # every feature must sum to the same totals (they describe the same rows)
    tg, th = int(G[1].sum()), int(H[1].sum())
    G[0, 255] += tg - int(G[0].sum())
    ref = _scan_reference(G, H, n, nb, mb, kg, kh, l1, l2, 0.0, 1e-3, 20)
    got = ops.split_scan(torch.from_numpy(G).cuda(), torch.from_numpy(H).cuda(), n, tg, th, nb, mb, kg, kh,
                         lambda_l1=l1, lambda_l2=l2)
    key, f, t, dl, a, b = ref[1]
    assert (got["feature"], got["threshold"], got["default_left"]) == (f, t, dl)
    assert got["left_g"] == a and got["left_h"] == b and got["total_cnt"] == n
    assert abs(got["gain"] - ref[0]) <= 1e-9 * abs(ref[0])


def test_device_binning_equals_host_binning():
    import torch
    rng = np.random.RandomState(0)
    X = rng.randn(30000, 37)
    X[:, 3] = rng.randint(0, 4, 30000)
    X[::9, 5] = np.nan
    X[:, 7] = 1.0
    for arr in (X, X.astype(np.float32), np.asfortranarray(X)):
        dc = lgb.Dataset(arr, label=np.zeros(30000), params=CPU).construct()
        dg = lgb.Dataset(arr, label=np.zeros(30000), params=GPU).construct()
        np.testing.assert_array_equal(dc.get_bins(), dg.get_bins())
    dt = lgb.Dataset(torch.from_numpy(X).cuda(), label=np.zeros(30000), params=GPU).construct()
    np.testing.assert_array_equal(dt.get_bins(), dc.get_bins())


def test_binary_cache_of_a_device_dataset(tmp_path):
    """Bins that only exist in HBM are fetched for the cache file; the loaded dataset trains the same model."""
    import torch
    X, y = make_regression(20000, 12, seed=21)
    d = lgb.Dataset(torch.from_numpy(X).cuda(), label=y, params=GPU)
    path = str(tmp_path / "d.bin")
    d.save_binary(path)
    e = lgb.Dataset(path, params=GPU)
    p = {**GPU, "num_leaves": 31}
    assert trees_of(lgb.train(p, d, 6)) == trees_of(lgb.train(p, e, 6))
    assert trees_of(lgb.train({**CPU, "num_leaves": 31}, lgb.Dataset(path, params=CPU), 6)) == trees_of(lgb.train(p, e, 6))


@pytest.mark.parametrize("n,sample,dtype", [(1500, 200000, np.float64), (50000, 200000, np.float32),
                                            (120000, 30000, np.float64)])
def test_device_bin_boundaries_equal_host_boundaries(n, sample, dtype):
    """Device sort + run-length encode + greedy cut gives bit-identical bounds to the host BinMapper."""
    import torch
    rng = np.random.RandomState(n)
    X = rng.randn(n, 11)
    X[:, 1] = rng.randint(-3, 4, n)                    # few distinct values, negative and zero
    X[:, 2] = np.round(rng.exponential(1.0, n), 1)     # heavy ties
    X[::7, 3] = np.nan
    X[:, 4] = -2.5                                     # constant
    X[:, 5] = rng.randint(0, 2, n) * 1e30
    X[:, 6] = np.where(rng.rand(n) < 0.9, 0.0, rng.randn(n))   # mostly one value
    X[::5, 7] = np.nan
    X[1::5, 7] = 3.0
    X[:, 8] = rng.randint(0, 1000, n)                  # more distinct values than bins
    X = X.astype(dtype)
    t = torch.from_numpy(X).cuda()
    base = {**GPU, "bin_construct_sample_cnt": sample, "max_bin": 255, "min_data_in_bin": 3}
    dd = lgb.Dataset(t, label=np.zeros(n), params={**base, "device_bin_find": True}).construct()
    dh = lgb.Dataset(t, label=np.zeros(n), params={**base, "device_bin_find": False}).construct()
    for f in range(X.shape[1]):
        a, b = dd.bin_info(f), dh.bin_info(f)
        assert a["num_bin"] == b["num_bin"] and a["missing_bin"] == b["missing_bin"], f
        np.testing.assert_array_equal(a["upper_bounds"], b["upper_bounds"])
    np.testing.assert_array_equal(dd.get_bins(), dh.get_bins())
    small = lgb.Dataset(t, label=np.zeros(n), params={**base, "max_bin": 16}).construct()
    assert max(small.bin_info(f)["num_bin"] for f in range(X.shape[1])) <= 16


# ----------------------------------------------------------------------------- engine parity
@pytest.mark.parametrize("params", [
    {"num_leaves": 31},
    {"num_leaves": 127, "min_data_in_leaf": 5, "lambda_l2": 0.5},
    {"num_leaves": 63, "max_depth": 5, "feature_fraction": 0.7},
    {"num_leaves": 31, "bagging_fraction": 0.6, "bagging_freq": 2},
    {"num_leaves": 15, "max_bin": 63, "lambda_l1": 0.1, "min_gain_to_split": 0.05},
    {"num_leaves": 63, "feature_fraction_bynode": 0.5, "min_data_in_leaf": 5},
])
def test_gpu_regression_trees_equal_cpu_trees(params):
    X, y = make_regression(40000, 12, seed=3)
    X[::13, 4] = np.nan
    w = np.where(X[:, 0] > 0, 1.5, 0.5)
    a = lgb.train({**CPU, **params}, lgb.Dataset(X, label=y, weight=w, params=CPU), 12)
    b = lgb.train({**GPU, **params}, lgb.Dataset(X, label=y, weight=w, params=GPU), 12)
    assert trees_of(a) == trees_of(b)


@pytest.mark.parametrize("n,F,leaves,extra", [
    (700, 6, 31, {}),
    (36700, 6, 127, {"bagging_fraction": 0.8, "bagging_freq": 4, "feature_fraction": 0.8}),   # a fold of the reference grid
    (40000, 12, 255, {"min_data_in_leaf": 3, "lambda_l2": 0.3}),            # two histogram passes at the top, 16 features' layout
    (65536, 28, 63, {"max_depth": 7, "extra_trees": True}),                 # the row limit, 32 features' layout
    (20000, 3, 255, {"feature_fraction_bynode": 0.7, "min_data_in_leaf": 1}),
    (5000, 9, 15, {"objective": "huber", "max_bin": 31}),
])
def test_whole_tree_kernel_grows_the_cpu_trees(n, F, leaves, extra):
    """Bags of up to 65536 rows are grown by ONE workgroup in one launch (the whole best-first loop on the device);
    the trees are the CPU oracle's and those of the batched growth rounds, and the leaf values renewed from the
    kernel's own 36-bit sums are the same as well."""
    X, y = make_regression(n, F, seed=n % 97)
    X[:, F - 1] = np.abs(np.round(X[:, F - 1] * 1.5))  # a handful of distinct values
    X[::11, 1] = np.nan
    cat = [F - 1] if F == 9 else "auto"
    p = {"num_leaves": leaves, "min_data_in_leaf": 5, "max_cat_to_onehot": 64, **extra}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU, categorical_feature=cat), 10)
    b = lgb.train({**GPU, **p, "whole_tree_max_rows": 65536}, lgb.Dataset(X, label=y, params=GPU, categorical_feature=cat), 10,
                  keep_training_booster=True)
    c = lgb.train({**GPU, **p, "whole_tree_max_rows": 0}, lgb.Dataset(X, label=y, params=GPU, categorical_feature=cat), 10,
                  keep_training_booster=True)
    assert b.profile()["chunk_iterations"] == 10 and c.profile()["chunk_iterations"] == 0 and c.grow_stats()["rounds"] > 1
    d = lgb.Booster({**GPU, **p, "whole_tree_max_rows": 65536}, lgb.Dataset(X, label=y, params=GPU, categorical_feature=cat))
    for _ in range(10):
        d.update()                      # one whole-tree launch per iteration, the host in between
    assert d.profile()["whole_trees"] == 10 and d.grow_stats()["rounds"] == 1 and trees_of(d) == trees_of(a)
    assert trees_of(a) == trees_of(b) == trees_of(c)
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True, device="cpu"), atol=1e-9)


@pytest.mark.parametrize("case", ["diamonds", "huber-nan-extra", "binary-bynode", "reg-sqrt", "poisson-weights"])
def test_boosting_chunks_on_the_device_equal_the_step_by_step_paths(case):
    """Small boosters run whole chunks of iterations on the device - one workgroup does gradients, bag, tree, leaf values,
    score updates and metric sums, with no host in between.  train() and cv() then give the CPU oracle's trees and,
    bit for bit, the metric values of the per-iteration device path; early stopping ends at the same iteration."""
    from lightgbm_amd.utils.datasets import make_diamonds_like
    if case == "diamonds":
        X, y = make_diamonds_like(45000, seed=0)
        w = None
        p = {"objective": "regression", "num_leaves": 63, "learning_rate": 0.1, "min_data_in_leaf": 20, "bagging_fraction": 0.8,
             "bagging_freq": 4, "feature_fraction": 0.8, "metric": ["l2", "l1"]}
    elif case == "huber-nan-extra":
        X, y = make_regression(20000, 12, seed=2)
        X[::9, 3] = np.nan
        w = None
        p = {"objective": "huber", "num_leaves": 31, "learning_rate": 0.2, "metric": "rmse", "max_depth": 6, "bagging_fraction": 0.5,
             "bagging_freq": 1, "extra_trees": True}
    elif case == "binary-bynode":
        X, y = make_regression(30000, 28, seed=5)
        y = (y > 0).astype(float)
        w = None
        p = {"objective": "binary", "num_leaves": 15, "learning_rate": 0.2, "metric": ["binary_logloss", "binary_error"],
             "feature_fraction_bynode": 0.6, "pos_bagging_fraction": 0.7, "neg_bagging_fraction": 0.9, "bagging_freq": 2}
    elif case == "reg-sqrt":
        X, y = make_regression(15000, 8, seed=4)
        y = np.exp(np.clip(y / 3, -3, 3)) * np.where(X[:, 2] > 1.2, -1.0, 1.0)
        w = None
        p = {"objective": "regression", "reg_sqrt": True, "num_leaves": 31, "learning_rate": 0.15, "metric": ["l2", "l1"],
             "min_data_in_leaf": 10}
    else:
        X, y = make_regression(9000, 5, seed=8)
        y = np.floor(np.exp(np.clip(y / 3, -3, 2)))
        w = 0.5 + np.random.RandomState(1).rand(len(y))
        p = {"objective": "poisson", "num_leaves": 20, "learning_rate": 0.1, "metric": "poisson", "min_data_in_leaf": 30}
    n_tr = len(y) * 4 // 5
    runs = {}
    for tag, extra in (("cpu", CPU), ("chunks", {**GPU, "whole_tree_max_rows": 65536}), ("steps", {**GPU, "whole_tree_max_rows": 0})):
        pp = {**extra, **p}
        dtr = lgb.Dataset(X[:n_tr], label=y[:n_tr], weight=None if w is None else w[:n_tr], params=pp)
        dva = dtr.create_valid(X[n_tr:], label=y[n_tr:], weight=None if w is None else w[n_tr:])
        ev = {}
        b = lgb.train(pp, dtr, 60, valid_sets=[dva], valid_names=["va"], early_stopping_rounds=4, evals_result=ev,
                      keep_training_booster=True)
        cvr = lgb.cv(pp, lgb.Dataset(X, label=y, weight=w, params=pp), 35, nfold=3, early_stopping_rounds=4, seed=2)
        runs[tag] = (b, ev, cvr)
    assert runs["chunks"][0].profile()["chunk_iterations"] > 0 and runs["steps"][0].profile()["chunk_iterations"] == 0
    assert trees_of(runs["cpu"][0]) == trees_of(runs["chunks"][0]) == trees_of(runs["steps"][0])
    assert runs["cpu"][0].best_iteration == runs["chunks"][0].best_iteration == runs["steps"][0].best_iteration
    assert runs["chunks"][1] == runs["steps"][1]                       # evaluation history, bit for bit
    assert runs["chunks"][2] == runs["steps"][2]                       # cross-validation history, bit for bit
    for k, v in runs["cpu"][2].items():
        np.testing.assert_allclose(runs["chunks"][2][k], v, rtol=1e-9)
    b = runs["chunks"][0]
    every = b.num_trees()   # (the scores hold the trees grown after the best iteration too)
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X[:n_tr], raw_score=True, device="cpu", num_iteration=every), atol=1e-9)
    np.testing.assert_allclose(b._inner_predict(0, raw=True), b.predict(X[n_tr:], raw_score=True, device="cpu", num_iteration=every), atol=1e-9)


def test_grid_rows_are_batched_into_single_launches_with_the_same_table():
    """Continuous batching: one thread keeps many grid rows alive and advances all their fold boosters with one
    native call - one kernel launch, one workgroup per booster; the table is the one-row-at-a-time table."""
    from lightgbm_amd.grid import expand_grid, grid_search_cv
    from lightgbm_amd.utils.datasets import make_diamonds_like
    X, y = make_diamonds_like(12000, seed=1)
    grid = expand_grid(iteration=None, score=None, learning_rate=[0.3, 0.1], num_leaves=[15, 63], min_data_in_leaf=[20],
                       feature_fraction=[0.8, 1], bagging_fraction=[0.7, 1], bagging_freq=4)
    kw = dict(nrounds=80, nfold=4, objective="regression", early_stopping_rounds=4, seed=3)
    one = grid_search_cv(grid, X=X, y=y, rows_in_flight=1, **kw)
    many = grid_search_cv(grid, X=X, y=y, rows_in_flight=11, **kw)
    cpu = grid_search_cv(grid, X=X, y=y, rows_in_flight=1, dataset_params={"device_type": "cpu"}, base_params={"device_type": "cpu"}, **kw)
    np.testing.assert_array_equal(one["score"].values, many["score"].values)
    np.testing.assert_array_equal(one["iteration"].values, many["iteration"].values)
    np.testing.assert_array_equal(one["iteration"].values, cpu["iteration"].values)
    np.testing.assert_allclose(one["score"].values, cpu["score"].values, rtol=1e-9)


@pytest.mark.parametrize("first", [0, 10, 20, 30])
def test_randomised_parity_sweep(first):
    """Random shapes (across the row-line, two-rounds-in-flight and tree-walk size gates), missing values,
    weights, subsets and parameter sets: the device engine grows the CPU oracle's trees, bit for bit."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import parity_sweep
    for seed in range(first, first + 10):
        same, case = parity_sweep.run_case(seed, 1_200_000)
        assert same, f"seed {seed}: {case}"


@pytest.mark.parametrize("first", [0, 12, 24])
def test_randomised_parity_sweep_full_precision(first):
    """The same sweep with ``use_quantized_grad=false``: the device's full-precision histograms (float gradients turned
    into 36-bit fixed point in every pass, two 64-bit LDS atomics per row and feature) give the CPU engine's
    full-precision trees, bit for bit - single-tile row lines and multi-tile records alike."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import parity_sweep
    for seed in range(first, first + 12):
        same, case = parity_sweep.run_case(seed, 1_200_000, full=True)
        assert same, f"seed {seed}: {case}"


@pytest.mark.parametrize("onehot", [4, 64])
def test_gpu_categorical_features_match_the_cpu_and_the_device_predictor_agrees(onehot):
    """onehot=4: the 40-category column is split on category SETS (histogram rows come back, the sorted search
    runs on the host, the partition and the tree walk read the sets); 64: one category against the rest, all
    on the device.  Either way the trees are the CPU engine's and the device predictor agrees with the host's."""
    rng = np.random.RandomState(3)
    n = 50000
    city = rng.randint(0, 40, size=n)
    effect = rng.randn(40) * 2
    X = np.c_[rng.randn(n), city.astype(float), rng.randn(n), rng.randint(0, 5, size=n).astype(float)]
    X[::19, 1] = np.nan
    y = np.where(np.isnan(X[:, 1]), 0.0, effect[city]) + 0.5 * X[:, 0] + 0.3 * (X[:, 3] == 2) + 0.2 * rng.randn(n)
    p = {"num_leaves": 63, "min_data_in_leaf": 10, "bagging_fraction": 0.8, "bagging_freq": 1, "max_cat_to_onehot": onehot}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, categorical_feature=[1, 3], params=CPU), 10)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, categorical_feature=[1, 3], params=GPU), 10, keep_training_booster=True)
    assert trees_of(a) == trees_of(b) and "cat_threshold=" in trees_of(b)
    set_sizes = [len(str(t).split("||")) for t in b.trees_to_dataframe().query("decision_type == '=='")["threshold"]]
    assert (max(set_sizes) > 3) == (onehot == 4)
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True, device="cpu"), atol=1e-9)
    np.testing.assert_allclose(b.predict(X, device="gpu"), b.predict(X, device="cpu"), atol=1e-12)
    Xf = X.astype(np.float32)
    np.testing.assert_allclose(b.predict(Xf, device="gpu"), b.predict(Xf, device="cpu"), atol=1e-12)


def test_index_checking_build_catches_a_bad_row_and_passes_clean_runs():
    """The -DLGA_DEBUG build of the HIP core (LGA_DEBUG_LIB=1): a row id beyond the dataset handed to the
    histogram kernel is reported as an error instead of faulting the GPU, and ordinary training (with
    bagging, missing values, out-of-bag tree walks) trips no check and still matches the CPU engine."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import lightgbm_amd as lgb
        from lightgbm_amd import _lib, ops
        assert _lib.is_hip() and str(_lib.PKG / "_lga_hip_debug.so")
        import parity_sweep
        for seed in (0, 2, 4, 7, 12):
            same, case = parity_sweep.run_case(seed, 600000)
            assert same, case
        N = 5000
        bins = torch.randint(0, 256, (1, N, 32), dtype=torch.uint8, device="cuda")
        packed = torch.ones(N, dtype=torch.int64, device="cuda")
        rows = torch.arange(1000, dtype=torch.int32, device="cuda")
        g, h, head = ops.hist_build(bins, rows, packed, 28)
        assert int(head[0]) == 1000
        rows[17] = N + 5
        try:
            ops.hist_build(bins, rows, packed, 28)
        except _lib.LightGBMError as e:
            assert "row id outside the dataset" in str(e), e
            print("CAUGHT", e)
        else:
            raise SystemExit("the bad row id went unnoticed")
        g, h, head = ops.hist_build(bins, torch.arange(1000, dtype=torch.int32, device="cuda"), packed, 28)
        assert int(head[0]) == 1000     # the device is fine afterwards
        print("OK")
    """) % (os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "tools"))
    env = dict(os.environ, LGA_DEBUG_LIB="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "CAUGHT" in r.stdout and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("budget", ["0", "1"])
def test_the_page_locked_block_cache_is_a_matter_of_time_only(budget):
    """Released boosters leave their page-locked host blocks in a process-wide cache up to a byte budget
    (LGA_HOST_CACHE_MB, read once per process).  With no budget every release frees its blocks at once, with 1 MiB the
    cache overflows after a few boosters: waves of boosters that come and go - a grid search's pattern - train the
    same models either way."""
    import json, os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, json, numpy as np
        sys.path.insert(0, %r)
        import lightgbm_amd as lgb
        rng = np.random.RandomState(3)
        X = rng.randn(20000, 6); y = X[:, 0] * X[:, 1] + np.sin(X[:, 2]) + 0.1 * rng.randn(20000)
        p = {"objective": "regression", "num_leaves": 31, "verbose": -1, "device_type": "gpu", "bagging_fraction": 0.8,
             "bagging_freq": 2, "min_data_in_leaf": 20}
        d = lgb.Dataset(X, label=y, params=p, free_raw_data=False)
        out = []
        for wave in range(3):     # every cv creates 5 fold boosters and drops them
            r = lgb.cv({**p, "learning_rate": 0.1 + 0.05 * wave}, d, num_boost_round=12, nfold=5, stratified=False, seed=1)
            out.append(r[sorted(r)[0]])
        b = lgb.train(p, d, 5)
        out.append(b.predict(X[:50], device="cpu").tolist())
        print("RESULT" + json.dumps(out))
    """) % (os.path.join(os.path.dirname(__file__), ".."),)

    def run(env):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=170)
        assert r.returncode == 0 and "RESULT" in r.stdout, r.stdout + r.stderr
        return json.loads(r.stdout[r.stdout.index("RESULT") + 6:])
    base = {k: v for k, v in os.environ.items() if k != "LGA_HOST_CACHE_MB"}
    assert run(dict(base, LGA_HOST_CACHE_MB=budget)) == run(base)


@pytest.mark.parametrize("lines,flight", [("0", "1"), ("1", "1"), ("1", "2"), ("0", "2"), ("1", "4")])
def test_row_line_layout_grows_the_same_trees(lines, flight, monkeypatch):
    """The large-data paths - bin records interleaved with the packed gradients, two (or four: large leaves then get
    rounds of their own and rounds are replayed as they finish) growth rounds in flight on as many streams - against
    the CPU oracle (both are size-gated, so they are forced here)."""
    monkeypatch.setenv("LGA_ROW_LINES", lines)
    monkeypatch.setenv("LGA_ROUNDS_IN_FLIGHT", flight)
    monkeypatch.setenv("LGA_OWN_ROUND_ROWS", "9000")   # (only read with more than two rounds in flight)
    X, y = make_regression(50001, 9, seed=11)     # row count not divisible by three
    X[::17, 2] = np.nan
    p = {"num_leaves": 63, "min_data_in_leaf": 10, "bagging_fraction": 0.7, "bagging_freq": 1}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 8)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params=GPU), 8)
    assert trees_of(a) == trees_of(b)


def assert_same_splits(a, b):
    """Same trees up to the last bits of the values (initial scores summed in another order): every structural line
    of the model text equal, the value lines close."""
    la, lb = trees_of(a).split("\n"), trees_of(b).split("\n")
    assert len(la) == len(lb)
    approx = ("leaf_value=", "internal_value=", "leaf_weight=", "internal_weight=", "split_gain=")
    for x, z in zip(la, lb):
        if x.startswith(approx):
            np.testing.assert_allclose(np.array(x.split("=")[1].split(), dtype=float), np.array(z.split("=")[1].split(), dtype=float),
                                       rtol=1e-9, atol=1e-10)
        else:
            assert x == z


def test_large_data_paths_match_the_cpu_engine():
    """2^20 rows and more take paths the small tests never reach by themselves: row lines, two rounds in flight, large
    leaves as sides / move / dense child histogram launches, leaf ids and exact leaf sums from one tree walk.  Same
    splits as the CPU engine; leaf values agree to the last bits (the initial score is summed in another order)."""
    X, y = make_regression((1 << 20) + 12345, 12, seed=3)
    p = {"num_leaves": 63, "min_data_in_leaf": 20}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 4)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params=GPU), 4)
    assert_same_splits(a, b)


def test_gradients_taken_with_the_score_update_change_nothing(monkeypatch):
    """Large data: the pass that adds a finished tree to the scores also takes the next iteration's gradients; they are
    used only while the scores stay as they were (a rollback in between must trigger a fresh pass).  Same models with
    the fusion switched off."""
    X, y = make_higgs_like((1 << 20) + 777, seed=9, device="numpy")
    w = np.where(X[:, 1] > 0, 2.0, 0.5)
    p = {**GPU, "objective": "binary", "num_leaves": 31, "min_sum_hessian_in_leaf": 10.0}

    def run():
        b = lgb.Booster(p, lgb.Dataset(X, label=y, weight=w, params=GPU))
        for _ in range(3):
            b.update()
        b.rollback_one_iter()
        for _ in range(2):
            b.update()
        return trees_of(b)
    monkeypatch.setenv("LGA_FUSE_GRADIENTS", "0")
    plain = run()
    monkeypatch.setenv("LGA_FUSE_GRADIENTS", "1")
    fused = run()
    assert plain == fused


def test_dense_histogram_units_cover_children_the_search_underestimated():
    """The grid of a large leaf's dense histogram units is sized from the child's ESTIMATED row count (hessian share):
    with weights spread over six orders of magnitude the estimate is off by far, and the units have to loop over
    what the child really holds.  Same trees as the CPU engine."""
    X, y = make_regression(120000, 8, seed=21)
    rng = np.random.RandomState(0)
    w = np.where(rng.rand(len(y)) < 0.02, 1e3, 1e-3)      # 2 % of the rows carry nearly all the hessian
    p = {"num_leaves": 31, "min_data_in_leaf": 50, "min_sum_hessian_in_leaf": 0.0}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, weight=w, params=CPU), 6)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, weight=w, params=GPU), 6)
    assert_same_splits(a, b)      # (the weighted mean the scores start from differs in the twelfth digit)


def test_gpu_forced_splits_equal_cpu(tmp_path):
    """forcedsplits_filename on the device engine: the forced nodes' row counts are not known ahead of time (the
    dense histogram units are sized from a placeholder and loop over what the child really has)."""
    import json
    X, y = make_regression(60000, 8, seed=13)
    path = tmp_path / "forced.json"
    path.write_text(json.dumps({"feature": 2, "threshold": 0.9, "left": {"feature": 0, "threshold": -1.2},
                                "right": {"feature": 5, "threshold": 0.0}}))
    p = {"num_leaves": 31, "min_data_in_leaf": 10, "forcedsplits_filename": str(path)}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 5)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params=GPU), 5)
    assert trees_of(a) == trees_of(b)
    assert a.dump_model()["tree_info"][0]["tree_structure"]["split_feature"] == 2


def test_gpu_balanced_bagging_draws_the_cpu_bag():
    X, y = make_higgs_like(80000, seed=5, device="numpy")
    y = (y > 0) & (X[:, 0] > 0.3)
    p = {"objective": "binary", "num_leaves": 31, "bagging_freq": 1, "pos_bagging_fraction": 0.9, "neg_bagging_fraction": 0.2,
         "boost_from_average": False}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y.astype(float), params=CPU), 5)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y.astype(float), params=GPU), 5, keep_training_booster=True)
    sa, sb = a.model_to_string(), b.model_to_string()
    assert sa[sa.index("Tree=0"):sa.index("Tree=1")] == sb[sb.index("Tree=0"):sb.index("Tree=1")]
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True, device="cpu"), atol=1e-9)


def test_gpu_binary_matches_cpu_quality_and_first_tree():
    X, y = make_higgs_like(60000, seed=2, device="numpy")
    p = {"objective": "binary", "num_leaves": 63, "metric": ["auc", "binary_logloss"], "min_sum_hessian_in_leaf": 10.0}
    ec, eg = {}, {}
    dc = lgb.Dataset(X, label=y, params=CPU); dg = lgb.Dataset(X, label=y, params=GPU)
    a = lgb.train({**CPU, **p}, dc, 30, valid_sets=[dc], evals_result=ec)
    b = lgb.train({**GPU, **p}, dg, 30, valid_sets=[dg], evals_result=eg)
    sa, sb = a.model_to_string(), b.model_to_string()
    assert sa[sa.index("Tree=0"):sa.index("Tree=1")] == sb[sb.index("Tree=0"):sb.index("Tree=1")]
    # exp() differs in the last bits between libm and the device, so later trees may differ in ties only
    assert abs(ec["training"]["auc"][-1] - eg["training"]["auc"][-1]) < 1e-3
    assert abs(ec["training"]["binary_logloss"][-1] - eg["training"]["binary_logloss"][-1]) < 1e-3


def test_gpu_auc_against_the_full_precision_cpu_learner_over_several_data_draws():
    """Held-out AUC on a Higgs-like task over three data draws.  The device engine with full-precision histograms
    (``use_quantized_grad=false``) takes the CPU full-precision learner's split decisions: the AUCs agree to 1e-9 on
    every draw.  The 16-bit learner's MEAN difference stays inside 0.001 (single pairs of two boosting runs drift by
    several 1e-4 at this size, so one pair says little).  The record at the benchmark's own size - 10.5 M rows, 500
    trees, five draws - is profiles/r3/auc_parity_500trees.txt (bench/auc_parity_multi.py)."""
    from sklearn.metrics import roc_auc_score
    diffs = []
    for seed in (21, 22, 23):
        X, y = make_higgs_like(250000, seed=seed, device="numpy")
        tr, te = slice(0, 200000), slice(200000, None)
        p = {"objective": "binary", "num_leaves": 127, "learning_rate": 0.1, "min_sum_hessian_in_leaf": 20.0, "verbose": -1}
        auc = {}
        for name, dev, quant in (("gpu q16", "gpu", True), ("gpu full", "gpu", False), ("cpu full", "cpu", False)):
            q = {**p, "device_type": dev, "use_quantized_grad": quant}
            b = lgb.train(q, lgb.Dataset(X[tr], label=y[tr], params=q), 80)
            auc[name] = roc_auc_score(y[te], b.predict(X[te], device="cpu"))
        assert abs(auc["gpu full"] - auc["cpu full"]) < 1e-9, (seed, auc)
        diffs.append(auc["gpu q16"] - auc["cpu full"])
    assert abs(np.mean(diffs)) <= 0.001 and max(abs(d) for d in diffs) <= 0.003, diffs


@pytest.mark.parametrize("case", ["walk", "bagging"])
def test_full_precision_leaf_values_need_no_renewal(case, monkeypatch):
    """With full-precision histograms a node's histogram totals ARE the 36-bit sums over its rows, so the device engine
    skips the renewal pass (and the host round trip it costs): forcing the pass (LGA_RENEW_ALWAYS=1) gives the same
    model, and so does the CPU engine, which always renews.  'walk': enough rows for the tree-walk score update."""
    rng = np.random.RandomState(23)
    n = 1_060_000 if case == "walk" else 120_000
    X = rng.randn(n, 6).astype(np.float32)
    y = (X[:, 0] + 0.5 * X[:, 1] * X[:, 2] + 0.3 * rng.randn(n) > 0).astype(np.float32)
    p = {"objective": "binary", "num_leaves": 31, "min_data_in_leaf": 20, "learning_rate": 0.2, "use_quantized_grad": False,
         "lambda_l2": 0.5, "verbose": -1}
    if case == "bagging":
        p.update(bagging_fraction=0.7, bagging_freq=1)

    def fit(dev):
        q = {**p, "device_type": dev}
        return lgb.train(q, lgb.Dataset(X, label=y, params=q), 4)
    monkeypatch.delenv("LGA_RENEW_ALWAYS", raising=False)
    skipped = fit("gpu")
    monkeypatch.setenv("LGA_RENEW_ALWAYS", "1")
    renewed = fit("gpu")
    monkeypatch.delenv("LGA_RENEW_ALWAYS", raising=False)
    assert trees_of(skipped) == trees_of(renewed)
    assert trees_of(skipped) == trees_of(fit("cpu"))


@pytest.mark.parametrize("case", ["plain", "bagging", "goss", "weights_init", "custom", "stochastic", "multiclass", "full_dart"])
def test_locality_order_changes_nothing_but_the_memory_layout(case, monkeypatch):
    """The device engine's private locality order of the rows (on by default from 2^20 rows; forced here at 150 k):
    the trees are the CPU engine's - which knows no such order - bit for bit, with bagging, GOSS and stochastic
    rounding (their random streams stay keyed by the caller's row numbers), weights and initial scores (reordered on
    the way in), custom gradients (scattered on the way in), multiclass score vectors and dart's tree replays; the
    training scores come back in the caller's row order."""
    rng = np.random.RandomState(17)
    n = 150000
    X = rng.randn(n, 9)
    X[:, 8] = rng.randint(0, 4, size=n)
    y = X[:, 0] * 1.5 + np.abs(X[:, 1]) - 0.7 * X[:, 2] * (X[:, 0] > 0) + 0.3 * (X[:, 8] == 2) + 0.1 * rng.randn(n)
    p = {"num_leaves": 63, "min_data_in_leaf": 5, "boost_from_average": False, "learning_rate": 0.2}
    kw, dkw, rounds = {}, {}, 5
    if case == "bagging":
        p.update(bagging_fraction=0.6, bagging_freq=1, feature_fraction=0.8)
    elif case == "goss":
        p.update(data_sample_strategy="goss", top_rate=0.2, other_rate=0.1, learning_rate=0.5)
    elif case == "weights_init":
        dkw = {"weight": 0.5 + rng.rand(n), "init_score": 0.3 * X[:, 3]}
        p.update(objective="huber")
    elif case == "custom":
        kw = {"fobj": lambda preds, d: (preds - d.get_label() + 0.1 * np.sign(preds), np.full(len(preds), 1.25))}
    elif case == "stochastic":
        p.update(stochastic_rounding=True, seed=5, bagging_fraction=0.8, bagging_freq=2)
    elif case == "multiclass":
        y = (np.digitize(y, np.quantile(y, [0.33, 0.66]))).astype(float)
        p.update(objective="multiclass", num_class=3)
        rounds = 3
    elif case == "full_dart":
        p.update(use_quantized_grad=False, boosting="dart", drop_rate=0.5, skip_drop=0.0)
        rounds = 7

    def fit(dev):
        base = {"device_type": dev, "verbose": -1}
        d = lgb.Dataset(X, label=y, params=base, **dkw)
        b = lgb.train({**base, **p}, d, rounds, keep_training_booster=True, **kw)
        return b
    a = fit("cpu")
    monkeypatch.setenv("LGA_LOCALITY_ORDER", "1")
    b = fit("gpu")
    monkeypatch.setenv("LGA_LOCALITY_ORDER", "0")
    c = fit("gpu")
    assert trees_of(b) == trees_of(c)                       # the order changes nothing ...
    if case != "multiclass":                                # (softmax goes through exp(): last-bit differences vs libm)
        assert trees_of(a) == trees_of(b)                   # ... and the trees are the CPU engine's
    raw_b, raw_c = b._inner_predict(-1, raw=True), c._inner_predict(-1, raw=True)
    np.testing.assert_allclose(raw_b, raw_c, rtol=0, atol=1e-12)       # scores in the caller's row order
    if case in ("plain", "bagging", "stochastic"):
        np.testing.assert_allclose(raw_b, b.predict(X, raw_score=True, device="cpu"), atol=1e-9)


@pytest.mark.parametrize("width,precision", [(12, "full"), (12, "q16"), (70, "full")])
def test_launch_shapes_and_pool_size_change_nothing_but_the_time(width, precision, monkeypatch):
    """How a round is cut into workgroups is a matter of speed only: the split search with one workgroup per child (every
    wave takes its features in turn, the next one's cells requested ahead) or with as many feature blocks as there are,
    histogram units of the smallest and of the largest size, and a histogram pool with room for 2 x or 8 x
    (num_leaves - 1) splits (a tree that runs out of slots finishes one leaf per round) all grow the CPU engine's trees.
    300 k rows: the row-line rounds of single-tile data; 70 features: the feature-tile rounds."""
    rng = np.random.RandomState(23)
    n = 300000 if width <= 32 else 60000
    X = rng.randn(n, width)
    X[:, 3] = rng.randint(0, 3, size=n)
    y = (X[:, 0] + 0.8 * X[:, 1] * X[:, 2] - 0.5 * np.abs(X[:, 4]) + 0.4 * (X[:, 3] == 1) + 0.5 * rng.randn(n) > 0).astype(float)
    p = {"objective": "binary", "num_leaves": 127, "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 20.0,
         "learning_rate": 0.2, "use_quantized_grad": precision == "q16", "verbose": -1}

    def fit(dev):
        base = {"device_type": dev, "verbose": -1}
        return trees_of(lgb.train({**p, **base}, lgb.Dataset(X, label=y, params={**p, **base}), 4))
    want = fit("cpu")
    for env in ({"LGA_SCAN_WGS": "1"}, {"LGA_SCAN_WGS": "100000"}, {"LGA_TARGET_UNITS": "1"}, {"LGA_TARGET_UNITS": "4096"},
                {"LGA_POOL_FACTOR": "2"}, {"LGA_POOL_FACTOR": "8", "LGA_SPEC_FLOOR": "0.0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = fit("gpu")
        for k in env:
            monkeypatch.delenv(k)
        assert got == want, env


@pytest.mark.parametrize("objective", ["multiclass", "multiclassova"])
def test_gpu_multiclass_matches_cpu(objective):
    """num_class score / gradient vectors on the device: the first iteration (scores all zero, so the
    gradients hold no device-specific exp() rounding) grows the CPU's trees; after that the metrics agree."""
    rng = np.random.RandomState(5)
    X = rng.randn(30000, 8)
    y = (X @ rng.randn(8, 4) + 0.4 * rng.randn(30000, 4)).argmax(axis=1).astype(np.float64)
    p = {"objective": objective, "num_class": 4, "num_leaves": 31, "boost_from_average": False,
         "metric": ["multi_logloss", "multi_error"]}
    ec, eg = {}, {}
    dc = lgb.Dataset(X, label=y, params=CPU); dg = lgb.Dataset(X, label=y, params=GPU)
    a = lgb.train({**CPU, **p}, dc, 20, valid_sets=[dc], evals_result=ec)
    b = lgb.train({**GPU, **p}, dg, 20, valid_sets=[dg], evals_result=eg, keep_training_booster=True)
    sa, sb = a.model_to_string(), b.model_to_string()
    assert sa[sa.index("Tree=0"):sa.index("Tree=4")] == sb[sb.index("Tree=0"):sb.index("Tree=4")]
    for m in ("multi_logloss", "multi_error"):
        assert abs(ec["training"][m][-1] - eg["training"][m][-1]) < 2e-3
    assert b.num_trees() == 80 and b.predict(X[:9]).shape == (9, 4)
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True), atol=1e-9)
    b.rollback_one_iter()
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True), atol=1e-9)


def test_gpu_multiclass_per_class_initial_scores():
    """[rows, classes] initial scores (continued training): the device adds class k's slice to class k's
    scores, for the training data and for a validation set of its own."""
    rng = np.random.RandomState(8)
    X = rng.randn(20000, 6)
    y = (X @ rng.randn(6, 3)).argmax(axis=1).astype(np.float64)
    p = {"objective": "multiclass", "num_class": 3, "num_leaves": 15, "metric": "multi_logloss"}
    first = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 3)
    out = {}
    for name, dev in (("cpu", CPU), ("gpu", GPU)):
        d = lgb.Dataset(X, label=y, params=dev, free_raw_data=False)
        dv = lgb.Dataset(X[:5000], label=y[:5000], reference=d, params=dev, free_raw_data=False)
        ev = {}
        b = lgb.train({**dev, **p}, d, 3, init_model=first, valid_sets=[d, dv], evals_result=ev, keep_training_booster=True)
        assert b.num_trees() == 18
        np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True), atol=1e-9)
        np.testing.assert_allclose(b._inner_predict(0, raw=True), b.predict(X[:5000], raw_score=True), atol=1e-9)
        out[name] = ev
    for k in ("training", "valid_1"):
        assert abs(out["cpu"][k]["multi_logloss"][-1] - out["gpu"][k]["multi_logloss"][-1]) < 1e-3


@pytest.mark.parametrize("objective", ["multiclass", "multiclassova"])
def test_device_predictor_serves_multiclass_models(objective):
    """One pass per class over device-resident rows plus the softmax / sigmoid kernel, against the host predictor:
    normal and raw output, staged prediction, float32 / float64 rows, host and device matrices, rf averaging."""
    import torch
    rng = np.random.RandomState(8)
    X = rng.randn(20000, 12)
    y = (X[:, :5] @ rng.randn(5, 5) + 0.5 * rng.randn(20000, 5)).argmax(axis=1).astype(np.float64)
    for extra in ({}, {"boosting": "rf", "bagging_fraction": 0.6, "bagging_freq": 1}):
        b = lgb.train({**GPU, "objective": objective, "num_class": 5, "num_leaves": 15, **extra}, lgb.Dataset(X, label=y), 12)
        for kw in ({}, {"raw_score": True}, {"num_iteration": 5}, {"start_iteration": 3, "num_iteration": 4}):
            want = b.predict(X, device="cpu", **kw)
            assert want.shape == (20000, 5)
            np.testing.assert_allclose(b.predict(X, device="gpu", **kw), want, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(b.predict(torch.from_numpy(X).cuda(), **kw), want, rtol=1e-12, atol=1e-12)
        X32 = X.astype(np.float32)
        np.testing.assert_allclose(b.predict(X32, device="gpu"), b.predict(X32, device="cpu"), rtol=1e-12, atol=1e-12)


def _rank_data(seed, nq, lo, hi, F=10):
    rng = np.random.RandomState(seed)
    sizes = rng.randint(lo, hi, size=nq)
    n = int(sizes.sum())
    X = rng.randn(n, F)
    y = np.clip(np.round(X[:, 0] + np.sin(X[:, 1]) + 0.4 * rng.randn(n) + 1.5), 0, 4)
    return X, y, sizes, 0.5 + rng.rand(n), rng.randn(n)


@pytest.mark.parametrize("objective", ["lambdarank", "rank_xendcg"])
@pytest.mark.parametrize("lo,hi,nq", [(1, 60, 400), (200, 900, 12), (1500, 4096, 3)])
def test_device_ranking_gradients_match_the_host(objective, lo, hi, nq):
    """One workgroup per query (LDS sort + per-document pair sums) against the host's per-query loop, from
    random scores (tied ones included), with weights, on a row subset (dataset row -> root row indirection)
    and for every documents-per-thread instantiation of the kernel.  Same formula, same summation order:
    what is left is the last bits of the device's exp() / log2()."""
    X, y, sizes, w, init = _rank_data(11, nq, lo, hi)
    init = np.round(init, 1)                      # plenty of tied scores
    keep_q = np.arange(nq) % 3 != 1               # a subset made of whole queries
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    rows = np.concatenate([np.arange(bounds[q], bounds[q + 1]) for q in np.flatnonzero(keep_q)])
    p = {"objective": objective, "min_data_in_leaf": 1, "lambdarank_truncation_level": 20, "sigmoid": 1.2}
    p.update({"metric": ["ndcg", "map"], "eval_at": [1, 3, 10, 50]})
    grads, metrics = [], []
    for dev in (CPU, GPU):
        d = lgb.Dataset(X, label=y, weight=w, group=sizes, init_score=init, params=dev).construct().subset(rows)
        b = lgb.Booster({**dev, **p}, d)
        b._train_data_name = "training"
        metrics.append(b.eval_train())         # at the initial scores, which both engines hold exactly
        b.update()
        grads.append(b.get_gradients())
    # the metric kernel (same sort, additions in the host's order) reproduces the host's values bit for bit
    assert len(metrics[0]) == 8 and metrics[0] == metrics[1]
    for host, device in zip(*grads):
        assert np.abs(host).max() > 0
        np.testing.assert_allclose(device, host, rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("objective", ["lambdarank", "rank_xendcg"])
def test_gpu_ranking_matches_cpu(objective, monkeypatch):
    """Device gradients: the first tree (all scores equal, so no exp() rounding enters) is the CPU's and the
    metrics stay together.  Host gradients under the device engine (LGA_RANK_ON_HOST): every tree is the CPU's."""
    X, y, sizes, w, _ = _rank_data(2, 1500, 3, 60)
    ntr = int(sizes[:1200].sum())
    p = {"objective": objective, "num_leaves": 31, "min_data_in_leaf": 10, "eval_at": [1, 5], "metric": ["ndcg", "map"],
         "bagging_fraction": 0.7, "bagging_freq": 1}

    def run(dev):
        dtr = lgb.Dataset(X[:ntr], label=y[:ntr], weight=w[:ntr], group=sizes[:1200], params=dev)
        dva = lgb.Dataset(X[ntr:], label=y[ntr:], group=sizes[1200:], reference=dtr, params=dev)
        ev = {}
        b = lgb.train({**dev, **p}, dtr, 15, valid_sets=[dva], valid_names=["va"], evals_result=ev)
        return trees_of(b), ev["va"]
    cpu_trees, cpu_ev = run(CPU)
    gpu_trees, gpu_ev = run(GPU)
    assert cpu_trees[:cpu_trees.index("Tree=1")] == gpu_trees[:gpu_trees.index("Tree=1")]
    for m in cpu_ev:
        assert abs(cpu_ev[m][-1] - gpu_ev[m][-1]) < 0.01
    assert gpu_ev["ndcg@5"][-1] > gpu_ev["ndcg@5"][0]
    monkeypatch.setenv("LGA_RANK_ON_HOST", "1")
    host_trees, host_ev = run(GPU)
    assert host_trees == cpu_trees and host_ev == cpu_ev


@pytest.mark.parametrize("leaves,n", [(4, 5000), (31, 70000)])
def test_device_auc_equals_host_auc(leaves, n):
    """Sort + scan AUC on the device against the host's sorted formulation, heavy ties included.
    A regression objective keeps CPU and GPU trees (hence scores) bit-identical."""
    rng = np.random.RandomState(leaves)
    X = rng.randn(n, 6)
    y = (X[:, 0] + 0.5 * X[:, 1] * X[:, 2] + 0.7 * rng.randn(n) > 0).astype(np.float64)
    p = {"objective": "regression", "num_leaves": leaves, "metric": ["auc"], "min_data_in_leaf": 5}
    idx = np.sort(rng.choice(n, n // 3, replace=False))
    out = {}
    for name, dev in (("cpu", CPU), ("gpu", GPU)):
        full = lgb.Dataset(X, label=y, params=dev)
        ev = {}
        lgb.train({**dev, **p}, full, 6, valid_sets=[full, full.subset(idx)], valid_names=["all", "part"],
                  evals_result=ev)
        out[name] = ev
    for k in ("all", "part"):
        assert out["cpu"][k]["auc"] == out["gpu"][k]["auc"], k
    assert 0.6 < out["gpu"]["all"]["auc"][-1] < 1.0


def test_gpu_predictor_matches_host_predictor():
    import torch
    X, y = make_regression(30000, 20, seed=5)
    X[::7, 2] = np.nan
    b = lgb.train({**GPU, "num_leaves": 255, "min_data_in_leaf": 3}, lgb.Dataset(X, label=y, params=GPU), 60)
    host = b.predict(X, device="cpu")
    np.testing.assert_allclose(b.predict(X, device="gpu"), host, rtol=0, atol=1e-12)
    np.testing.assert_allclose(b.predict(torch.from_numpy(X).cuda()), host, rtol=0, atol=1e-12)
    np.testing.assert_allclose(b.predict(X.astype(np.float32), device="gpu"), b.predict(X.astype(np.float32), device="cpu"),
                               rtol=0, atol=1e-12)
    np.testing.assert_allclose(b.predict(X, device="gpu", num_iteration=7), b.predict(X, device="cpu", num_iteration=7),
                               atol=1e-12)
    yb = (y > 0).astype(float)
    bb = lgb.train({**GPU, "objective": "binary"}, lgb.Dataset(X, label=yb, params=GPU), 10)
    np.testing.assert_allclose(bb.predict(X, device="gpu"), bb.predict(X, device="cpu"), atol=1e-12)
    rf = lgb.train({**GPU, "boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1},
                   lgb.Dataset(X, label=y, params=GPU), 10)
    np.testing.assert_allclose(rf.predict(X, device="gpu"), rf.predict(X, device="cpu"), atol=1e-12)


@pytest.mark.parametrize("cols,force", [(12, ""), (12, "rows64"), (70, ""), (260, ""), (700, "")])
def test_every_device_predictor_kernel_matches_the_host(cols, force, monkeypatch):
    """The device predictor picks its kernel by the width of the matrix: 256 rows per workgroup in LDS (12 columns), 64
    rows in LDS with one tree of the tile per wave (70 and 260 columns; forced for 12), feature values in global memory
    (700 columns).  Each adds a row's trees in order, so all equal the host predictor; NaNs, a categorical column searched
    over category sets, a 1-leaf tree (constant first iteration of a tiny learning rate), float32 and float64 matrices
    and a row count that leaves the last workgroup part empty."""
    rng = np.random.RandomState(cols)
    n = 9000 + 37
    X = rng.randn(n, cols)
    X[:, 1] = rng.randint(0, 40, size=n)
    X[rng.rand(n) < 0.05, 2] = np.nan
    y = X[:, 0] + 0.5 * X[:, cols - 1] * X[:, 3] + 0.3 * (X[:, 1] % 5 == 2) + 0.2 * rng.randn(n)
    p = {**GPU, "num_leaves": 63, "min_data_in_leaf": 5, "learning_rate": 0.2, "max_cat_to_onehot": 4,
         "min_gain_to_split": 0.0}
    b = lgb.train(p, lgb.Dataset(X, label=y, params=p, categorical_feature=[1]), 37)
    if force:
        monkeypatch.setenv("LGA_PRED_KERNEL", force)
    for M in (X, X.astype(np.float32)):
        host = b.predict(M, device="cpu")
        np.testing.assert_allclose(b.predict(M, device="gpu"), host, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b.predict(M[:65], device="gpu", num_iteration=5), b.predict(M[:65], device="cpu", num_iteration=5),
                                   rtol=0, atol=1e-12)
    yb = (y > np.median(y)).astype(float)
    bb = lgb.train({**p, "objective": "binary", "min_sum_hessian_in_leaf": 1e9}, lgb.Dataset(X, label=yb, params=p), 3)   # 1-leaf trees
    np.testing.assert_allclose(bb.predict(X, device="gpu"), bb.predict(X, device="cpu"), rtol=0, atol=1e-12)


def test_gpu_cv_and_rstyle_grid_row():
    from lightgbm_amd.rstyle import lgb_cv, lgb_Dataset
    X, y = make_diamonds_like(20000, seed=4)
    kw = dict(nrounds=200, nfold=5, objective="regression", eval="rmse", showsd=False, early_stopping_rounds=5,
              verbose=0, seed=3)
    g = lgb_cv({"learning_rate": 0.1}, lgb_Dataset(X, label=y, params=GPU), **kw)
    c = lgb_cv({"learning_rate": 0.1}, lgb_Dataset(X, label=y, params=CPU), **kw)
    assert g.best_iter == c.best_iter and abs(g.best_score - c.best_score) < 1e-9


def test_concurrent_fold_streams_give_the_same_cv():
    """Fold boosters on their own host threads / HIP streams: same numbers as one after the other."""
    X, y = make_diamonds_like(20000, seed=4)
    d = lgb.Dataset(X, label=y, params=GPU)
    p = {**GPU, "objective": "regression", "metric": "rmse", "learning_rate": 0.2, "num_leaves": 31}
    a = lgb.cv(p, d, 40, nfold=5, early_stopping_rounds=5, seed=3, concurrent_folds=False)
    b = lgb.cv(p, d, 40, nfold=5, early_stopping_rounds=5, seed=3, concurrent_folds=True)
    assert a["rmse-mean"] == b["rmse-mean"] and a["rmse-stdv"] == b["rmse-stdv"]


def test_bagging_over_many_partition_chunks():
    """The bagging partition spans hundreds of 2048-row chunks: later chunks must still see the parent range
    after the first one has published the bag / out-of-bag ranges."""
    X, y = make_regression(1_500_000, 4, seed=12)
    # no boost-from-average: the label mean of 1.5 M rows depends on the summation order in its last bits
    p = {"num_leaves": 15, "bagging_fraction": 0.5, "bagging_freq": 1, "min_data_in_leaf": 50,
         "boost_from_average": False}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 3)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params=GPU), 3)
    assert trees_of(a) == trees_of(b)


def test_concurrent_folds_with_bagging_and_feature_fraction():
    """A grid-search row of the reference (r/gridsearchCV.R:11-18): bagging + feature sampling, five fold
    boosters on their own streams."""
    X, y = make_diamonds_like(43000, seed=5)
    d = lgb.Dataset(X, label=y, params=GPU)
    p = {**GPU, "objective": "regression", "metric": "rmse", "learning_rate": 0.1, "num_leaves": 31,
         "min_data_in_leaf": 20, "feature_fraction": 0.8, "bagging_fraction": 0.6, "bagging_freq": 4}
    a = lgb.cv(p, d, 60, nfold=5, early_stopping_rounds=5, seed=3, concurrent_folds=False)
    b = lgb.cv(p, d, 60, nfold=5, early_stopping_rounds=5, seed=3, concurrent_folds=True)
    assert a["rmse-mean"] == b["rmse-mean"] and a["rmse-stdv"] == b["rmse-stdv"]


def test_gpu_rf_and_rollback_and_custom_objective():
    X, y = make_regression(20000, 8, seed=6)
    p = {"boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1, "feature_fraction": 0.8, "num_leaves": 63}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params=CPU), 8)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params=GPU), 8)
    assert trees_of(a) == trees_of(b)
    bst = lgb.train(GPU, lgb.Dataset(X, label=y, params=GPU), 6, keep_training_booster=True)
    bst.rollback_one_iter()
    np.testing.assert_allclose(bst._inner_predict(-1, raw=True), bst.predict(X), atol=1e-9)
    fobj = lambda preds, data: (preds - data.get_label(), np.ones_like(preds))
    c = lgb.train({**GPU, "boost_from_average": False}, lgb.Dataset(X, label=y, params=GPU), 5, fobj=fobj)
    d = lgb.train({**GPU, "boost_from_average": False}, lgb.Dataset(X, label=y, params=GPU), 5)
    np.testing.assert_allclose(c.predict(X), d.predict(X), atol=1e-9)


def test_many_features_use_several_tiles():
    X, y = make_regression(8000, 150, seed=7)
    a = lgb.train({**CPU, "num_leaves": 31}, lgb.Dataset(X, label=y, params=CPU), 5)
    b = lgb.train({**GPU, "num_leaves": 31}, lgb.Dataset(X, label=y, params=GPU), 5)
    assert trees_of(a) == trees_of(b)


# ----------------------------------------------------------------------------- distributed code paths on one GPU
def _run_ranks(P, fn):
    from lightgbm_amd.parallel import DeviceLoopbackComm
    comms = DeviceLoopbackComm.make_group(P)
    out, err = [None] * P, [None] * P

    def work(r):
        try:
            out[r] = fn(r, comms[r])
        except BaseException as e:   # noqa
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in err:
        if e is not None:
            raise e
    [c.close() for c in comms]
    return out


@pytest.mark.parametrize("exchange", ["allreduce", "reduce_scatter"])
@pytest.mark.parametrize("P,flight,lines", [(2, "1", "0"), (4, "1", "1"), (2, "2", "1"), (4, "2", "0"), (8, "2", "1")])
def test_gpu_data_parallel_equals_serial(P, flight, lines, exchange, monkeypatch):
    """Both exchanges of the data-parallel learner: whole histograms by all-reduce (the default for narrow data: one
    collective per round, every rank searches every feature) and reduce-scatter by feature + winner all-gather."""
    monkeypatch.setenv("LGA_DP_EXCHANGE", exchange)
    monkeypatch.setenv("LGA_ROUNDS_IN_FLIGHT", flight)   # "2": overlapped rounds
    monkeypatch.setenv("LGA_ROW_LINES", lines)           # "1": the row-line layout large shards use
    X, y = make_regression(40000, 10, seed=8)
    params = {**GPU, "num_leaves": 63}
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X[:20000], params=GPU), 10)
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X[:20000], params=GPU)
        return lgb.train({**params, "tree_learner": "data"}, d, 10, comm=comm).model_to_string()
    for s in _run_ranks(P, rank):
        assert trees_of(s) == trees_of(serial)


def _cpu_voting_ranks(P, X, y, sample, params, rounds):
    """The CPU engine's voting-parallel learner over P thread ranks (the reference for the device's)."""
    from lightgbm_amd.parallel import ThreadComm
    comms = ThreadComm.make_group(P)
    bounds = np.linspace(0, len(y), P + 1).astype(int)
    out, err = [None] * P, [None] * P

    def work(r):
        try:
            sl = slice(bounds[r], bounds[r + 1])
            d = lgb.Dataset(X[sl], label=y[sl], bin_sample=sample, params=CPU)
            out[r] = lgb.train({**CPU, **params}, d, rounds, comm=comms[r]).model_to_string()
        except BaseException as e:   # noqa
            err[r] = e
            comms[r]._hub.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    [c.close() for c in comms]
    return out


@pytest.mark.parametrize("P,F,top_k,lines,precision", [(2, 10, 5, "0", "q16"), (3, 28, 3, "1", "q16"), (4, 70, 4, "0", "full"),
                                                     (3, 28, 20, "1", "full")])
def test_gpu_voting_parallel_equals_the_cpu_engines_vote(P, F, top_k, lines, precision, monkeypatch):
    """PV-Tree on the device (local search, top_k votes per rank and child, the 2 x top_k most voted features' rows
    summed, search in the summed rows): every rank grows the trees the CPU engine's voting learner grows over the same
    shards - with few votes, over row lines and over feature tiles, in both histogram precisions - and with
    2 x top_k >= F the serial trees."""
    monkeypatch.setenv("LGA_ROW_LINES", lines)
    X, y = make_regression(36000, F, seed=13)
    sample = X[:20000]
    q = {"use_quantized_grad": precision == "q16"}
    params = {"num_leaves": 31, "min_data_in_leaf": 20, "tree_learner": "voting", "top_k": top_k, **q}
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=sample, params=GPU)
        return lgb.train({**GPU, **params}, d, 6, comm=comm).model_to_string()
    dev = _run_ranks(P, rank)
    cpu = _cpu_voting_ranks(P, X, y, sample, params, 6)
    for s in dev:
        assert trees_of(s) == trees_of(cpu[0])
    serial = lgb.train({**GPU, **q, "num_leaves": 31, "min_data_in_leaf": 20},
                       lgb.Dataset(X, label=y, bin_sample=sample, params=GPU), 6)
    if 2 * top_k >= F:
        assert trees_of(dev[0]) == trees_of(serial)


def test_gpu_voting_parallel_is_a_real_approximation_and_still_the_cpu_engines():
    """Few votes on data whose features all carry a little signal (Higgs-shaped, 0/1 labels as a regression target so
    that no exp() separates the engines): the trees differ from the serial learner's - the vote does restrict the
    search - and equal the CPU engine's voting trees rank for rank."""
    from lightgbm_amd.utils.datasets import make_higgs_like
    X, y = make_higgs_like(30000, seed=5, device="numpy")
    y = y.astype(np.float64)
    sample = X[:20000]
    P = 3
    params = {"num_leaves": 31, "min_data_in_leaf": 20, "tree_learner": "voting", "top_k": 2, "learning_rate": 0.2}
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=sample, params=GPU)
        return lgb.train({**GPU, **params}, d, 8, comm=comm).model_to_string()
    dev = _run_ranks(P, rank)
    cpu = _cpu_voting_ranks(P, X, y, sample, params, 8)
    assert trees_of(dev[0]) == trees_of(dev[1]) == trees_of(dev[2]) == trees_of(cpu[0])
    serial = lgb.train({**GPU, "num_leaves": 31, "min_data_in_leaf": 20, "learning_rate": 0.2},
                       lgb.Dataset(X, label=y, bin_sample=sample, params=GPU), 8)
    assert trees_of(dev[0]) != trees_of(serial)


@pytest.mark.parametrize("P,F,leaves", [(2, 40, 31), (3, 70, 63), (8, 5, 15)])
def test_gpu_data_parallel_over_feature_tiles_and_idle_owners(P, F, leaves, monkeypatch):
    """Reduce-scatter by feature: several feature tiles (records + gradient array instead of row lines), a
    feature count the ranks do not divide, and more ranks than features (ranks that own nothing to search).
    The loop-back communicator reduces ONLY the owner's block, like RCCL."""
    monkeypatch.setenv("LGA_DP_EXCHANGE", "reduce_scatter")
    X, y = make_regression(30000, F, seed=11)
    params = {**GPU, "num_leaves": leaves}
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=X[:20000], params=GPU), 6)
    bounds = np.linspace(0, len(y), P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=X[:20000], params=GPU)
        return lgb.train({**params, "tree_learner": "data"}, d, 6, comm=comm).model_to_string()
    for s in _run_ranks(P, rank):
        assert trees_of(s) == trees_of(serial)


@pytest.mark.parametrize("exchange,precision", [("allreduce", "q16"), ("reduce_scatter", "q16"), ("allreduce", "full"), ("reduce_scatter", "full")])
def test_gpu_data_parallel_with_shards_that_cross_the_size_gates_by_themselves(exchange, precision, monkeypatch):
    """Four ranks with 330 k rows each (no size gate forced through the environment): every shard ALONE takes the
    row-line kernels with multi-unit leaves, dense units and the tree-walk score update of its rank's engine, in both
    exchanges and both histogram precisions; every rank ends with the serial learner's trees."""
    monkeypatch.setenv("LGA_DP_EXCHANGE", exchange)
    P, n = 4, 1_320_000
    X, y = make_regression(n, 28, seed=31)
    q = {"use_quantized_grad": precision == "q16"}
    params = {**GPU, **q, "num_leaves": 127, "min_data_in_leaf": 20}
    sample = X[:: max(1, n // 200000)]
    serial = lgb.train(params, lgb.Dataset(X, label=y, bin_sample=sample, params=GPU), 5)
    bounds = np.linspace(0, n, P + 1).astype(int)

    def rank(r, comm):
        sl = slice(bounds[r], bounds[r + 1])
        d = lgb.Dataset(X[sl], label=y[sl], bin_sample=sample, params=GPU)
        return lgb.train({**params, "tree_learner": "data"}, d, 5, comm=comm).model_to_string()
    for s in _run_ranks(P, rank):
        assert trees_of(s) == trees_of(serial)


@pytest.mark.parametrize("precision", ["q16", "full"])
def test_gpu_feature_parallel_at_the_epsilon_width_with_eight_ranks(precision):
    """BASELINE config 5's learner: 2000 dense features (63 feature tiles) dealt over EIGHT feature-parallel ranks
    (loop-back communicator on one GPU; every rank holds all rows, histograms and searches its tiles, 32-byte winners
    are all-gathered) at the GPU-benchmark settings (255 leaves, 255 bins, min_sum_hessian 100): the serial trees."""
    from lightgbm_amd.utils.datasets import make_epsilon_like
    X, y = make_epsilon_like(40000, 2000, seed=3)
    X, y = X.numpy(), y.numpy()
    params = {**GPU, "objective": "binary", "num_leaves": 255, "max_bin": 255, "min_data_in_leaf": 1,
              "min_sum_hessian_in_leaf": 100.0, "use_quantized_grad": precision == "q16"}
    # (the first iteration's gradients involve no device-specific exp() rounding either way: all ranks are this device)
    serial = lgb.train(params, lgb.Dataset(X, label=y, params=params), 3)
    res = _run_ranks(8, lambda r, c: lgb.train({**params, "tree_learner": "feature"},
                                               lgb.Dataset(X, label=y, params=params), 3, comm=c).model_to_string())
    for s in res:
        assert trees_of(s) == trees_of(serial)


def test_gpu_feature_parallel_equals_serial():
    X, y = make_regression(10000, 100, seed=9)
    params = {**GPU, "num_leaves": 31}
    serial = lgb.train(params, lgb.Dataset(X, label=y, params=GPU), 6)
    res = _run_ranks(2, lambda r, c: lgb.train({**params, "tree_learner": "feature"},
                                               lgb.Dataset(X, label=y, params=GPU), 6, comm=c).model_to_string())
    for s in res:
        assert trees_of(s) == trees_of(serial)


def test_rccl_single_rank_communicator(tmp_path, monkeypatch):
    # RCCL itself needs one GPU per rank; with one GPU we can still bring a 1-rank communicator up and train through it
    import os, socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
    dist.init_process_group("gloo")
    try:
        from lightgbm_amd.parallel import RcclComm
        comm = RcclComm()
        X, y = make_regression(5000, 6, seed=10)
        a = lgb.train({**GPU, "tree_learner": "data"}, lgb.Dataset(X, label=y, params=GPU), 3, comm=comm)
        b = lgb.train(GPU, lgb.Dataset(X, label=y, params=GPU), 3)
        assert trees_of(a) == trees_of(b)
        # a communicator of one rank normally means the serial learner; LGA_DP_KEEP_SINGLE_RANK keeps the distributed
        # one, so the collectives (RCCL, one rank), the search hand-over behind them and the collective stream of two
        # rounds in flight all run - with both exchanges, on data large enough for the row-line kernels
        X, y = make_regression(300000, 28, seed=11)
        p = {**GPU, "num_leaves": 63, "min_data_in_leaf": 5}
        want = trees_of(lgb.train(p, lgb.Dataset(X, label=y, params=GPU), 4))
        monkeypatch.setenv("LGA_DP_KEEP_SINGLE_RANK", "1")
        for exchange, flight in (("allreduce", "1"), ("reduce_scatter", "1"), ("allreduce", "2"), ("reduce_scatter", "2")):
            monkeypatch.setenv("LGA_DP_EXCHANGE", exchange)
            monkeypatch.setenv("LGA_ROUNDS_IN_FLIGHT", flight)
            got = lgb.train({**p, "tree_learner": "data"}, lgb.Dataset(X, label=y, params=GPU), 4, comm=comm)
            assert trees_of(got) == want, (exchange, flight)
        comm.close()
    finally:
        dist.destroy_process_group()


def test_module_level_objects_do_not_crash_the_interpreter_at_exit(tmp_path):
    """A script's global boosters / datasets are still alive when the interpreter shuts down; their device
    memory is released by an exit hook while the HIP runtime is whole (a destructor running during module
    teardown used to end the process with a segmentation fault)."""
    import os, subprocess, sys, textwrap
    script = tmp_path / "globals.py"
    script.write_text(textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np, lightgbm_amd as lgb
        rng = np.random.RandomState(0)
        X = rng.randn(3000, 5); y = X[:, 0] + 0.1 * rng.randn(3000)
        p = {"device_type": "gpu", "verbose": -1}
        d = lgb.Dataset(X, label=y, params=p)
        b = lgb.train(p, d, 3, valid_sets=[d], verbose_eval=False, keep_training_booster=True)
        cvb = lgb.cv(p, lgb.Dataset(X, label=y, params=p), 3, nfold=2, return_cvbooster=True)
        print("END_OF_SCRIPT", flush=True)
    """) % os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240)
    assert "END_OF_SCRIPT" in r.stdout and r.returncode == 0, (r.returncode, r.stderr[-1500:])


def test_fallback_communicator_works_under_an_nccl_group(tmp_path):
    """default_comm() falls back to the process group's own collectives when the native RCCL communicator
    cannot be created; under an nccl group those take device tensors, so host buffers make a detour."""
    import os, socket, subprocess, sys, textwrap
    script = tmp_path / "fallback.py"
    script.write_text(textwrap.dedent("""
        import sys
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from lightgbm_amd.parallel.comm import TorchComm
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        c = TorchComm()
        v = np.arange(10, dtype=np.int64)
        c.allreduce(v, "sum")
        f = np.array([1.5, -2.0], dtype=np.float32)
        c.allreduce(f, "max")
        out = np.zeros(4, dtype=np.uint8)
        c.allgather(np.array([1, 2, 3, 4], dtype=np.uint8), out)
        assert v.tolist() == list(range(10)) and f.tolist() == [1.5, -2.0] and out.tolist() == [1, 2, 3, 4]
        c.close()
        dist.destroy_process_group()
        print("FALLBACK_OK")
    """) % os.path.join(os.path.dirname(__file__), ".."))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"}
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("P", [2, 4, 8])
def test_rccl_data_parallel_over_several_gpus(P, tmp_path):
    """One process per GPU, histograms summed by RCCL: every rank ends with the serial learner's trees.
    Needs P visible GPUs (skipped on a smaller box); started through torch.distributed.run on 127.0.0.1."""
    import os, socket, subprocess, sys, textwrap
    import torch
    if torch.cuda.device_count() < P:
        pytest.skip(f"needs {P} GPUs, {torch.cuda.device_count()} visible")
    script = tmp_path / "dp_rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        import lightgbm_amd as lgb
        from lightgbm_amd.parallel import RcclComm
        from lightgbm_amd.utils.datasets import make_regression
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group("nccl")
        GPU = {"device_type": "gpu", "verbose": -1}
        X, y = make_regression(60000, 12, seed=3)
        bounds = np.linspace(0, len(y), world + 1).astype(int)
        sl = slice(bounds[rank], bounds[rank + 1])
        comm = RcclComm()
        p = {**GPU, "num_leaves": 31, "tree_learner": "data"}
        b = lgb.train(p, lgb.Dataset(X[sl], label=y[sl], bin_sample=X, params=GPU), 8, comm=comm)
        text = b.model_to_string()
        cut = text[text.index("Tree=0"):text.index("end of trees")]
        if rank == 0:
            s = lgb.train({**GPU, "num_leaves": 31}, lgb.Dataset(X, label=y, bin_sample=X, params=GPU), 8).model_to_string()
            assert cut == s[s.index("Tree=0"):s.index("end of trees")], "data-parallel trees differ from serial"
        gathered = [None] * world
        dist.all_gather_object(gathered, cut)
        assert all(g == gathered[0] for g in gathered), "ranks disagree"
        comm.close()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """) % os.path.join(os.path.dirname(__file__), ".."))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and r.stdout.count("ok") == P, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("P", [2, 4])
def test_multi_process_data_parallel_sharing_one_gpu(P, tmp_path):
    """The multi-rank flow of a real run - one process per rank, torch.distributed rendezvous, bin boundaries
    agreed over the ranks, one histogram all-reduce per growth round - rehearsed on ONE GPU: the ranks share the
    device and reduce through gloo (the device engine stages its buffers through pinned memory).  Every rank must
    end with the serial learner's trees.  Also runs bench.py's multi-rank path end to end."""
    import json, os, socket, subprocess, sys, textwrap
    root = os.path.join(os.path.dirname(__file__), "..")
    script = tmp_path / "dp_rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        import lightgbm_amd as lgb
        from lightgbm_amd.parallel import default_comm, TorchComm
        from lightgbm_amd.parallel.data_parallel import gather_bin_sample
        from lightgbm_amd.utils.datasets import make_regression
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        GPU = {"device_type": "gpu", "verbose": -1}
        X, y = make_regression(40000, 12, seed=3)
        bounds = np.linspace(0, len(y), world + 1).astype(int)
        sl = slice(bounds[rank], bounds[rank + 1])
        p = {**GPU, "num_leaves": 31, "tree_learner": "data"}
        comm = default_comm()
        assert isinstance(comm, TorchComm)
        sample = gather_bin_sample(X[sl], p)
        b = lgb.train(p, lgb.Dataset(X[sl], label=y[sl], params=p, bin_sample=sample), 6, comm=comm)
        text = b.model_to_string()
        cut = text[text.index("Tree=0"):text.index("end of trees")]
        if rank == 0:
            q = {**GPU, "num_leaves": 31}
            s = lgb.train(q, lgb.Dataset(X, label=y, params=q, bin_sample=sample), 6).model_to_string()
            assert cut == s[s.index("Tree=0"):s.index("end of trees")], "data-parallel trees differ from serial"
        gathered = [None] * world
        dist.all_gather_object(gathered, cut)
        assert all(g == gathered[0] for g in gathered), "ranks disagree"
        comm.close()
        dist.destroy_process_group()
        print("rank", rank, "ok", flush=True)
    """) % root)

    def launch(args, env=None):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), *args],
                              capture_output=True, text=True, timeout=170, env=env)
    r = launch([str(script)])
    assert r.returncode == 0 and r.stdout.count("ok") == P, r.stdout[-2000:] + r.stderr[-2000:]
    r = launch([os.path.join(root, "bench.py"), "--gpus", str(P), "--rows", "300000", "--steps", "4", "--warmup", "1",
                "--no-full-run"],
               env=dict(os.environ, LGA_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 alone reports
    out = json.loads(lines[0])
    assert out["n_gpus"] == P and out["steps"] == 4 and out["value"] > 0 and out["config"]["rows"] == 300000
    if P == 2:
        # without a launcher bench.py starts its rank processes itself (before touching the GPU)
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "300000", "--steps", "3",
                            "--warmup", "1", "--no-full-run", "--precision", "q16"],
                           capture_output=True, text=True, timeout=170, env=dict(env, LGA_BENCH_BACKEND="gloo"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["config"]["comm_ranks"] == 2 and out["config"]["precision"] == "q16"
        assert "MI355X" in out["config"]["device"]


def test_grid_search_with_worker_processes_on_the_gpu(tmp_path):
    """The task-parallel grid search (one worker process per listed device, rows pulled from a queue, durable
    checkpoint): here three workers share GPU 0.  Same table as the in-process search; a finished checkpoint is
    not recomputed."""
    from lightgbm_amd.grid import expand_grid, grid_search_cv
    X, y = make_diamonds_like(6000, seed=8)
    grid = expand_grid(iteration=None, score=None, learning_rate=[0.3, 0.1], num_leaves=[7, 31], min_data_in_leaf=[20],
                       bagging_fraction=[0.8, 1.0], bagging_freq=[4])
    kw = dict(nrounds=60, nfold=3, objective="regression", early_stopping_rounds=3, seed=5, dataset_params=GPU,
              base_params=GPU)
    here = grid_search_cv(grid, X=X, y=y, **kw)
    ck = str(tmp_path / "grid.jsonl")
    spread = grid_search_cv(grid, X=X, y=y, workers=[0, 0, 0], checkpoint=ck, **kw)
    assert spread["iteration"].tolist() == here["iteration"].tolist()
    np.testing.assert_allclose(spread["score"].to_numpy(), here["score"].to_numpy(), rtol=0, atol=1e-12)
    again = grid_search_cv(grid, X=X, y=y, workers=[0], checkpoint=ck, **kw)       # everything is in the checkpoint
    assert again["score"].tolist() == spread["score"].tolist()
    workers_used = {__import__("json").loads(ln)["worker"] for ln in open(ck)}
    assert workers_used <= {"0"} and len(open(ck).readlines()) == len(grid)


def test_forced_bins_and_reg_sqrt_on_the_device_engine(tmp_path):
    """forcedbins_filename sends the boundary search to the host finder (the bins are uploaded afterwards) and
    reg_sqrt is an objective kind of the shared header: the device engine grows the CPU engine's trees with both."""
    import json
    X, y = make_regression(30000, 6, seed=21)
    y = np.exp(np.clip(y / 3, -3, 3))
    path = tmp_path / "fb.json"
    path.write_text(json.dumps([{"feature": 1, "bin_upper_bound": [-0.5, 0.0, 0.5]}, {"feature": 4, "bin_upper_bound": [1.25]}]))
    p = {"objective": "regression", "reg_sqrt": True, "forcedbins_filename": str(path), "max_bin": 31, "num_leaves": 31,
         "min_data_in_leaf": 5, "metric": "l2"}
    a = lgb.train({**CPU, **p}, lgb.Dataset(X, label=y, params={**CPU, **p}), 8)
    b = lgb.train({**GPU, **p}, lgb.Dataset(X, label=y, params={**GPU, **p}), 8)
    assert trees_of(a) == trees_of(b)
    assert "objective=regression sqrt" in b.model_to_string()
    np.testing.assert_allclose(b.predict(X, device="gpu"), a.predict(X), rtol=1e-9, atol=1e-12)
