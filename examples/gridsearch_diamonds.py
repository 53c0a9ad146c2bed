"""r/gridsearchCV.R, line by line, on this framework.

The real `diamonds` table cannot be downloaded here, so a synthetic table of the same shape
(53 940 rows, 6 features of which 3 are small ordinal integers, log-price target) stands in; pass
``--csv diamonds.csv`` to run on the real data (columns carat, cut, color, clarity, depth, table, price).

    python examples/gridsearch_diamonds.py                 # quick: 12-point grid
    python examples/gridsearch_diamonds.py --full          # the reference's 108-point grid
    python examples/gridsearch_diamonds.py --workers 0,1,2,3,4,5,6,7   # one worker process per GPU
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lightgbm_amd.grid import expand_grid, grid_search_cv, rank_grid, refit_top      # noqa: E402
from lightgbm_amd.rstyle import as_list, lgb_cv, lgb_Dataset, lgb_train, predict      # noqa: E402
from lightgbm_amd.utils.datasets import make_diamonds_like                            # noqa: E402


def load_data(csv):
    if csv is None:
        return make_diamonds_like(53940, seed=0)
    import pandas as pd
    d = pd.read_csv(csv)                                                     # R:5-14 data preparation
    cut = {"Fair": 1, "Good": 2, "Very Good": 3, "Premium": 4, "Ideal": 5}
    color = {c: i + 1 for i, c in enumerate("DEFGHIJ")}
    clarity = {c: i + 1 for i, c in enumerate(["I1", "SI2", "SI1", "VS2", "VS1", "VVS2", "VVS1", "IF"])}
    X = np.column_stack([np.log(d["carat"]), d["cut"].map(cut), d["color"].map(color), d["clarity"].map(clarity),
                         d["depth"], d["table"]]).astype(np.float64)
    return X, np.log(d["price"].values.astype(np.float64))


def rmse(y, pred):                                                           # R:36-38
    return float(np.sqrt(np.mean((y - pred) ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default=None)
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--workers", default=None, help="comma separated GPU ids (or 'cpu,cpu') for worker processes")
    ap.add_argument("--checkpoint", default="paramGrid.jsonl")
    ap.add_argument("--m", type=int, default=5)
    ap.add_argument("--rows-in-flight", type=int, default=None,
                    help="grid rows one process evaluates concurrently (default: 4 on the GPU build)")
    ap.add_argument("--fresh", action="store_true", help="ignore an existing checkpoint instead of resuming from it")
    args = ap.parse_args()

    X, y = load_data(args.csv)
    rng = np.random.RandomState(3928272)                                      # R:20-24
    inn = rng.choice([False, True], size=len(y), replace=True, p=[0.15, 0.85])
    train = {"X": X[inn], "y": y[inn]}
    test = {"X": X[~inn], "y": y[~inn]}

    lin = np.linalg.lstsq(np.c_[train["X"], np.ones(inn.sum())], train["y"], rcond=None)[0]     # R:45-46
    print("linear model rmse      ", rmse(test["y"], np.c_[test["X"], np.ones((~inn).sum())] @ lin))

    dtrain = lgb_Dataset(train["X"], label=train["y"])                        # R:52-53
    params = {"learning_rate": 0.1}                                           # R:55
    t = time.time()
    fit_lgb = lgb_train(params=params, data=dtrain, nrounds=200, objective="regression", verbose=0)   # R:57-61
    print("lgb.train 200 rounds    %.2fs" % (time.time() - t))
    print("untuned rmse           ", rmse(test["y"], predict(fit_lgb, test["X"])))              # R:63-64

    t = time.time()
    fit_lgb2 = lgb_cv(params=params, data=dtrain, nrounds=1000, nfold=5, objective="regression", eval="rmse",
                      showsd=False, early_stopping_rounds=5, verbose=0)       # R:70-78
    info = as_list(fit_lgb2)                                                  # R:79
    print("lgb.cv                  %.2fs  best_iter=%d  rmse=%.5f" % (time.time() - t, info["best_iter"], -info["best_score"]))

    if args.full:                                                             # R:92-100
        grid = expand_grid(iteration=None, score=None, learning_rate=[0.1, 0.05, 0.01], num_leaves=[31, 63, 127],
                           min_data_in_leaf=[20, 40], feature_fraction=[0.8, 1], bagging_fraction=[0.6, 0.8, 1],
                           bagging_freq=4, nthread=4)
    else:
        grid = expand_grid(iteration=None, score=None, learning_rate=[0.1, 0.05], num_leaves=[31, 63, 127],
                           min_data_in_leaf=[20], feature_fraction=[1], bagging_fraction=[0.8, 1], bagging_freq=4,
                           nthread=4)
    print("grid rows              ", len(grid))                               # R:102
    workers = None
    if args.workers:
        workers = [w if w == "cpu" else int(w) for w in args.workers.split(",")]
    t = time.time()
    grid = grid_search_cv(grid, X=train["X"], y=train["y"], dtrain=None if workers else dtrain, nrounds=1000, nfold=5,
                          objective="regression", early_stopping_rounds=5, checkpoint=args.checkpoint,
                          workers=workers, verbose=True, resume=not args.fresh,
                          rows_in_flight=args.rows_in_flight)                 # R:104-119
    secs = time.time() - t
    iters = int(grid["iteration"].clip(lower=0).sum()) + 5 * len(grid)
    print("grid search             %.1fs  (%d rows, ~%d boosting iterations x 5 folds)" % (secs, len(grid), iters))
    ranked = rank_grid(grid)                                                  # R:122
    print(ranked.head(12).to_string())
    if args.full:
        # what the reference's own saved grid shows (paramGrid.RData): the best rows use every feature and the
        # smaller leaf minimum, and learning rates <= 0.05 beat 0.1 given enough rounds
        top12, top10 = ranked.head(12), ranked.head(10)
        print("trend: top-12 rows with feature_fraction = 1      : %d of 12" % int((top12["feature_fraction"] == 1).sum()))
        print("trend: top-10 rows with min_data_in_leaf = 20     : %d of 10" % int((top10["min_data_in_leaf"] == 20).sum()))
        best_by_lr = grid.groupby("learning_rate")["score"].max()
        print("trend: best score by learning rate                :", {float(k): round(float(v), 7) for k, v in best_by_lr.items()})
        print("trend: rank of the best learning_rate = 0.1 row   : %d" %
              (1 + int(np.flatnonzero(ranked["learning_rate"].values == 0.1)[0])))
    pred, _ = refit_top(ranked, args.m, dtrain, test["X"])                    # R:124-143
    print("mean of top-%d rmse     " % args.m, rmse(test["y"], pred))         # R:144


if __name__ == "__main__":
    main()
