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


def test_trends_of_the_reference_run():
    # NBR:801: lower learning rates reach better CV scores when given enough rounds
    X, y = make_regression(3000, 10, seed=3, noise=0.5)
    grid = expand_grid(iteration=None, score=None, learning_rate=[0.3, 0.05], num_leaves=[15], min_data_in_leaf=[20])
    g = grid_search_cv(grid, X=X, y=y, nrounds=400, nfold=3, early_stopping_rounds=5, seed=1,
                       dataset_params={"device_type": "cpu"})
    hi, lo = g.iloc[0], g.iloc[1]
    assert lo["score"] > hi["score"] and lo["iteration"] > hi["iteration"]
