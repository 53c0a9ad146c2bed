"""Random-forest / bagging mode vs boosting (bagging_boosting.ipynb)."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_curve_1d, make_regression

CPU = {"device_type": "cpu", "verbose": -1}


def test_rf_prediction_is_the_tree_average():
    X, y = make_regression(3000, 6, seed=0)
    p = {**CPU, "boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1, "feature_fraction": 0.8, "num_leaves": 31}
    b = lgb.train(p, lgb.Dataset(X, label=y), 20)
    s = b.model_to_string()
    assert "average_output" in s.split("feature_names")[0]
    single = np.column_stack([b.predict(X, start_iteration=i, num_iteration=1) for i in range(20)])
    np.testing.assert_allclose(b.predict(X), single.mean(axis=1), atol=1e-9)
    # the flag survives a round trip
    b2 = lgb.Booster(model_str=s)
    np.testing.assert_allclose(b2.predict(X), b.predict(X), atol=1e-12)
    # trees are independent fits of the label (not of residuals): each alone is a decent model
    assert all(np.corrcoef(single[:, i], y)[0, 1] > 0.6 for i in range(20))


def test_rf_eval_tracks_running_average():
    X, y = make_regression(3000, 6, seed=1)
    d = lgb.Dataset(X[:2000], label=y[:2000])
    v = d.create_valid(X[2000:], label=y[2000:])
    ev = {}
    p = {**CPU, "boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1, "metric": "rmse"}
    b = lgb.train(p, d, 15, valid_sets=[v], evals_result=ev)
    final = np.sqrt(np.mean((b.predict(X[2000:]) - y[2000:]) ** 2))
    assert abs(ev["valid_0"]["rmse"][-1] - final) < 1e-6   # labels are float32 inside the engine
    assert ev["valid_0"]["rmse"][-1] < ev["valid_0"]["rmse"][0]      # averaging more trees reduces variance


def test_bagging_vs_boosting_on_the_toy_curve():
    # NBB:118-137 (boosting, staged prediction) and NBB:194-212 (forest of 1 / 3 / 100 trees)
    xg, yg, xt, yt = make_curve_1d()
    d = lgb.Dataset(xt, label=yt)
    boost = lgb.train({**CPU, "learning_rate": 0.02, "num_leaves": 31, "max_depth": 6, "min_data_in_leaf": 5}, d, 500)
    err = [np.sqrt(np.mean((boost.predict(xg, num_iteration=k) - yg) ** 2)) for k in (1, 20, 50, 100, 300)]
    assert all(a > b for a, b in zip(err, err[1:]))                 # more trees -> closer to the curve
    assert err[-1] < 0.05
    rf_err = {}
    for n in (1, 3, 100):
        rf = lgb.train({**CPU, "boosting": "rf", "num_leaves": 20, "bagging_fraction": 0.632, "bagging_freq": 1,
                        "min_data_in_leaf": 5, "seed": 345, "bagging_seed": 345}, lgb.Dataset(xt, label=yt), n)
        rf_err[n] = np.sqrt(np.mean((rf.predict(xg) - yg) ** 2))
    assert rf_err[100] < rf_err[1] and rf_err[100] < 0.08
    # one boosted tree at eta 0.02 is still far from the curve; one deep bagged tree is already close
    assert rf_err[1] < err[0]


@pytest.mark.parametrize("freq", [1, 3])
def test_forest_grown_in_parts_equals_the_forest_grown_in_one_go(freq):
    """Tree-parallel rf: iteration k draws the same bag / feature subset wherever it is trained."""
    from lightgbm_amd.parallel import merge_forests, train_forest_part, tree_share
    X, y = make_regression(3000, 8, seed=12)
    p = {**CPU, "boosting": "rf", "num_leaves": 31, "bagging_fraction": 0.63, "bagging_freq": freq,
         "feature_fraction": 0.8, "min_data_in_leaf": 5}
    ds = lgb.Dataset(X, label=y, params=CPU)
    whole = lgb.train(p, ds, 14)
    parts = [train_forest_part(p, ds, tree_share(14, 3, r)) for r in range(3)]
    assert [b.num_trees() for b in parts] == [4, 5, 5]
    merged = merge_forests([b.model_to_string() for b in parts])
    assert merged.num_trees() == 14
    a, b = whole.model_to_string(), merged.model_to_string()
    assert a[a.index("Tree=0"):a.index("end of trees")] == b[b.index("Tree=0"):b.index("end of trees")]
    np.testing.assert_allclose(merged.predict(X), whole.predict(X), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("P", [2, 4])
def test_tree_parallel_forest_over_processes_equals_the_one_process_forest(P, tmp_path):
    """bagging_boosting.ipynb:11 - bagged trees "are fitted in parallel": P processes (gloo) each grow their share of
    the iterations on the full table, the parts are gathered and merged on every rank; the forest equals the one a
    single process grows, tree for tree, and so do its predictions.  Also runs the 1 M x 50 example's multi-rank
    path (rf tree-parallel, gbdt data-parallel) at a small size."""
    import os, socket, subprocess, sys, textwrap
    root = os.path.join(os.path.dirname(__file__), "..")
    script = tmp_path / "forest_rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np, torch.distributed as dist
        sys.path.insert(0, %r)
        import lightgbm_amd as lgb
        from lightgbm_amd.parallel import train_forest_tree_parallel
        from lightgbm_amd.utils.datasets import make_regression
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        CPU = {"device_type": "cpu", "verbose": -1}
        X, y = make_regression(4000, 8, seed=12)
        p = {**CPU, "boosting": "rf", "num_leaves": 31, "bagging_fraction": 0.63, "bagging_freq": 1,
             "feature_fraction": 0.8, "min_data_in_leaf": 5}
        merged = train_forest_tree_parallel(p, X, y, 13)
        assert merged.num_trees() == 13
        if rank == 0:
            whole = lgb.train(p, lgb.Dataset(X, label=y, params=CPU), 13)
            a, b = whole.model_to_string(), merged.model_to_string()
            assert a[a.index("Tree=0"):a.index("end of trees")] == b[b.index("Tree=0"):b.index("end of trees")]
            np.testing.assert_allclose(merged.predict(X), whole.predict(X), rtol=1e-12, atol=1e-12)
        texts = [None] * world
        dist.all_gather_object(texts, merged.model_to_string())
        assert all(t == texts[0] for t in texts)
        dist.destroy_process_group()
        print("rank", rank, "ok", flush=True)
    """) % root)

    def launch(args):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), *args],
                              capture_output=True, text=True, timeout=280)
    r = launch([str(script)])
    assert r.returncode == 0 and r.stdout.count("ok") == P, r.stdout[-2000:] + r.stderr[-2000:]
    if P == 2:
        r = launch([os.path.join(root, "examples", "rf_vs_gbdt_1m.py"), "--rows", "20000", "--features", "10", "--trees", "8",
                    "--leaves", "15", "--device", "cpu"])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "tree-parallel, 2 ranks x 4 trees" in r.stdout and "data-parallel, 2 ranks x 10000 rows" in r.stdout, r.stdout
