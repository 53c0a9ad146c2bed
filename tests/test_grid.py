"""Grid search plumbing of r/gridsearchCV.R on a small synthetic regression, on the CPU engine."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from lightgbm_amd.grid import (average_predictions, expand_grid, grid_search_cv, load_checkpoint, rank_grid, refit_top)
from lightgbm_amd.rstyle import lgb_Dataset
from lightgbm_amd.utils.datasets import make_regression

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_grid():
    return expand_grid(iteration=None, score=None, learning_rate=[0.1, 0.05, 0.01], num_leaves=[31, 63, 127],
                       min_data_in_leaf=[20, 40], feature_fraction=[0.8, 1], bagging_fraction=[0.6, 0.8, 1],
                       bagging_freq=4, nthread=4)


def test_expand_grid_matches_the_saved_reference_grid():
    g = reference_grid()
    assert len(g) == 108 and list(g.columns) == ["iteration", "score", "learning_rate", "num_leaves",
                                                 "min_data_in_leaf", "feature_fraction", "bagging_fraction",
                                                 "bagging_freq", "nthread"]
    # first factor varies fastest: rows 1-3 of paramGrid.RData differ only in learning_rate
    assert g["learning_rate"].tolist()[:4] == [0.1, 0.05, 0.01, 0.1]
    assert g["num_leaves"].tolist()[:4] == [31, 31, 31, 63]
    assert g.iloc[:3][["num_leaves", "min_data_in_leaf", "feature_fraction", "bagging_fraction"]].nunique().max() == 1
    assert g["iteration"].isna().all() and g["score"].isna().all()
    assert (g["bagging_freq"] == 4).all() and (g["nthread"] == 4).all()
    # rows 63 and 99 (1-based), the two best of the saved artifact
    r63, r99 = g.iloc[62], g.iloc[98]
    assert (r63.learning_rate, r63.num_leaves, r63.min_data_in_leaf, r63.feature_fraction, r63.bagging_fraction) == (0.01, 127, 20, 1, 0.8)
    assert (r99.learning_rate, r99.num_leaves, r99.min_data_in_leaf, r99.feature_fraction, r99.bagging_fraction) == (0.01, 127, 20, 1, 1)


def small_grid():
    return expand_grid(iteration=None, score=None, learning_rate=[0.3, 0.1], num_leaves=[7, 15],
                       min_data_in_leaf=[20], bagging_fraction=[0.7, 1], bagging_freq=2, nthread=1)


KW = dict(nrounds=60, nfold=3, objective="regression", early_stopping_rounds=3, seed=5,
          dataset_params={"device_type": "cpu"})


def test_grid_search_fills_table_ranks_and_refits(tmp_path):
    X, y = make_regression(3000, 10, seed=0, noise=0.5)
    Xte, yte = make_regression(1000, 10, seed=0, noise=0.5)[0], None
    ck = str(tmp_path / "paramGrid.jsonl")
    g = grid_search_cv(small_grid(), X=X[:2400], y=y[:2400], checkpoint=ck, **KW)
    assert g["score"].notna().all() and (g["score"] < 0).all() and (g["iteration"] >= 1).all()
    assert len(load_checkpoint(ck)) == 8
    ranked = rank_grid(g)
    assert ranked["score"].is_monotonic_decreasing
    d = lgb_Dataset(X[:2400], label=y[:2400], params={"device_type": "cpu"})
    mean, preds = refit_top(ranked, 3, d, X[2400:])
    assert len(preds) == 3
    np.testing.assert_allclose(mean, average_predictions(preds))
    rm = lambda p: np.sqrt(np.mean((p - y[2400:]) ** 2))
    assert rm(mean) <= max(rm(p) for p in preds) + 1e-12
    assert rm(mean) < 1.05 * min(rm(p) for p in preds)       # averaging the best few is about as good as the best


def test_resume_skips_finished_rows_and_reproduces_the_table(tmp_path):
    X, y = make_regression(2000, 8, seed=1)
    ck = str(tmp_path / "ck.jsonl")
    with pytest.raises(RuntimeError, match="injected"):
        grid_search_cv(small_grid(), X=X, y=y, checkpoint=ck, inject_fail_after=3, **KW)
    done = load_checkpoint(ck)
    assert len(done) == 3
    calls = []
    g2 = grid_search_cv(small_grid(), X=X, y=y, checkpoint=ck, progress=lambda a, b: calls.append(a), **KW)
    assert len(calls) == 5 and calls[-1] == 8               # only the 5 missing rows ran
    full = grid_search_cv(small_grid(), X=X, y=y, checkpoint=None, **KW)
    np.testing.assert_allclose(g2["score"].values, full["score"].values)
    np.testing.assert_array_equal(g2["iteration"].values, full["iteration"].values)
    # a torn last line (killed mid-write) is ignored
    with open(ck, "a") as f:
        f.write('{"row": 7, "hash": "x", "iter')
    assert len(load_checkpoint(ck)) == 8
    # changed search settings invalidate old rows
    g3 = grid_search_cv(small_grid(), X=X, y=y, checkpoint=ck, progress=lambda a, b: calls.append(("new", a)),
                        **{**KW, "nfold": 4})
    assert sum(1 for c in calls if isinstance(c, tuple)) == 8 and g3["score"].notna().all()


def test_killed_process_resumes_from_checkpoint(tmp_path):
    # the reference saves after every row "if lgb crashes" (R:118): kill the whole process mid-search
    ck = str(tmp_path / "ck.jsonl")
    script = f"""
import sys, os
sys.path.insert(0, {ROOT!r})
os.environ["LGA_FORCE_HOST"] = "1"
from lightgbm_amd.grid import grid_search_cv
from lightgbm_amd.utils.datasets import make_regression
from tests.test_grid import small_grid, KW
X, y = make_regression(2000, 8, seed=1)
def die(done, total):
    if done == 4:
        os._exit(9)
grid_search_cv(small_grid(), X=X, y=y, checkpoint={ck!r}, progress=die, **KW)
"""
    r = subprocess.run([sys.executable, "-c", script], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 9, r.stderr
    assert len(load_checkpoint(ck)) == 4
    X, y = make_regression(2000, 8, seed=1)
    g = grid_search_cv(small_grid(), X=X, y=y, checkpoint=ck, **KW)
    full = grid_search_cv(small_grid(), X=X, y=y, **KW)
    np.testing.assert_allclose(g["score"].values, full["score"].values)


def test_worker_processes_and_dead_worker_requeue(tmp_path):
    X, y = make_regression(1500, 6, seed=2)
    ck = str(tmp_path / "ck.jsonl")
    # worker 0 dies with a row in flight after finishing one; the row is handed back and the table completes
    g = grid_search_cv(small_grid(), X=X, y=y, checkpoint=ck, workers=["cpu", "cpu"], inject_fail_after=1, **KW)
    full = grid_search_cv(small_grid(), X=X, y=y, **KW)
    np.testing.assert_allclose(g["score"].values, full["score"].values)
    recs = load_checkpoint(ck)
    assert len(recs) == 8 and len({r["worker"] for r in recs.values()}) >= 1


def test_rows_in_flight_do_not_change_the_table(tmp_path):
    X, y = make_regression(2000, 8, seed=4)
    ck = str(tmp_path / "ck.jsonl")
    seq = grid_search_cv(small_grid(), X=X, y=y, rows_in_flight=1, **KW)
    par = grid_search_cv(small_grid(), X=X, y=y, rows_in_flight=3, checkpoint=ck, **KW)
    np.testing.assert_array_equal(par["score"].values, seq["score"].values)
    np.testing.assert_array_equal(par["iteration"].values, seq["iteration"].values)
    assert len(load_checkpoint(ck)) == 8
    # a failure on one thread stops the pulls, keeps what finished and surfaces in the caller
    ck2 = str(tmp_path / "ck2.jsonl")
    with pytest.raises(RuntimeError, match="injected"):
        grid_search_cv(small_grid(), X=X, y=y, rows_in_flight=3, checkpoint=ck2, inject_fail_after=4, **KW)
    assert 1 <= len(load_checkpoint(ck2)) <= 4
    again = grid_search_cv(small_grid(), X=X, y=y, rows_in_flight=2, checkpoint=ck2, **KW)
    np.testing.assert_array_equal(again["score"].values, seq["score"].values)
    # worker processes with two rows in flight each
    w = grid_search_cv(small_grid(), X=X, y=y, workers=["cpu", "cpu"], rows_in_flight=2, **KW)
    np.testing.assert_array_equal(w["score"].values, seq["score"].values)
    # continuous batching: one thread keeps three rows alive and advances all their fold boosters per native call
    ck3 = str(tmp_path / "ck3.jsonl")
    bat = grid_search_cv(small_grid(), X=X, y=y, rows_in_flight=3, batch_rows=True, lockstep_chunk=4, checkpoint=ck3, **KW)
    np.testing.assert_array_equal(bat["score"].values, seq["score"].values)
    np.testing.assert_array_equal(bat["iteration"].values, seq["iteration"].values)
    assert len(load_checkpoint(ck3)) == 8


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("static", [False, True])
def test_distributed_search_pulls_rows_dynamically_and_merges_rank_checkpoints(tmp_path, static):
    """Two gloo processes share the grid through the store counter (r/gridsearchCV.R:104-121 as a task queue):
    the table equals the in-process one, every row ran exactly once, and a restart - here with rows left behind
    by a run of a different world size - only computes what no rank's checkpoint holds."""
    X, y = make_regression(2000, 8, seed=6)
    full = grid_search_cv(small_grid(), X=X, y=y, **KW)
    ck = str(tmp_path / "ck.jsonl")
    # leftovers of an earlier 4-rank run: two finished rows in rank files, one row of a search with other settings
    first = grid_search_cv(small_grid().iloc[:2], X=X, y=y, checkpoint=ck + ".rank3", **KW)
    assert first["score"].notna().all()
    stale = grid_search_cv(small_grid().iloc[:3], X=X, y=y, checkpoint=ck + ".rank1", **{**KW, "nfold": 4})
    assert stale["score"].notna().all()
    out = str(tmp_path / "table_rank{}.json")
    script = f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
os.environ["LGA_FORCE_HOST"] = "1"
import torch.distributed as dist
dist.init_process_group("gloo")
from lightgbm_amd.grid import grid_search_cv_distributed
from lightgbm_amd.utils.datasets import make_regression
from tests.test_grid import small_grid, KW
X, y = make_regression(2000, 8, seed=6)
g = grid_search_cv_distributed(small_grid(), X, y, checkpoint={ck!r}, static_shards={static!r}, **KW)
r = dist.get_rank()
json.dump({{"iteration": g["iteration"].tolist(), "score": g["score"].tolist()}}, open({out!r}.format(r), "w"))
dist.destroy_process_group()
"""
    path = str(tmp_path / "worker.py")
    with open(path, "w") as f:
        f.write(script)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), path],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    t0, t1 = json.load(open(out.format(0))), json.load(open(out.format(1)))
    assert t0 == t1
    np.testing.assert_array_equal(np.array(t0["score"]), full["score"].values)
    np.testing.assert_array_equal(np.array(t0["iteration"]), full["iteration"].values)
    # this run computed the six rows nobody had, each exactly once
    new = {}
    for rank in (0, 1):
        recs = load_checkpoint(ck + f".rank{rank}")
        recs = {i: v for i, v in recs.items() if v["worker"].startswith("rank")}
        assert not (set(recs) & set(new))
        new.update(recs)
    assert sorted(new) == [2, 3, 4, 5, 6, 7]


def test_trends_of_the_reference_run():
    # NBR:801: lower learning rates reach better CV scores when given enough rounds
    X, y = make_regression(3000, 10, seed=3, noise=0.5)
    grid = expand_grid(iteration=None, score=None, learning_rate=[0.3, 0.05], num_leaves=[15], min_data_in_leaf=[20])
    g = grid_search_cv(grid, X=X, y=y, nrounds=400, nfold=3, early_stopping_rounds=5, seed=1,
                       dataset_params={"device_type": "cpu"})
    hi, lo = g.iloc[0], g.iloc[1]
    assert lo["score"] > hi["score"] and lo["iteration"] > hi["iteration"]
