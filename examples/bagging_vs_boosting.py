"""bagging_boosting.ipynb on this framework: boosting with staged prediction vs a bagged forest.

The notebook fits xgboost (boosting) and sklearn's RandomForestRegressor (bagging) to a noisy
1-D curve and overlays predictions; here both model families come from the same engine
(``boosting="gbdt"`` and ``boosting="rf"``).  Pass ``--plot out.png`` to save the figure
(needs matplotlib), otherwise the RMSE against the true curve is printed.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import lightgbm_amd as lgb                                   # noqa: E402
from lightgbm_amd.utils.datasets import make_curve_1d        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plot", default=None)
    args = ap.parse_args()
    x, y, xtrain, ytrain = make_curve_1d()                                              # NBB:67-78
    dtrain = lgb.Dataset(xtrain, label=ytrain, free_raw_data=False)                     # NBB:118
    params = {"objective": "regression", "metric": "rmse", "learning_rate": 0.02, "max_depth": 6,
              "num_leaves": 31, "min_data_in_leaf": 5, "verbose": -1}                   # NBB:121-126
    t = time.time()
    cv = lgb.cv(params, dtrain, num_boost_round=1000, early_stopping_rounds=50, nfold=5)     # NBB:128
    print("cv: best round %d, rmse %.4f (%.2fs)" % (len(cv["rmse-mean"]), cv["rmse-mean"][-1], time.time() - t))
    t = time.time()
    model = lgb.train(params, dtrain, 500)                                              # NBB:131
    print("train 500 rounds: %.2fs" % (time.time() - t))
    staged = {k: model.predict(x, num_iteration=k) for k in (1, 20, 50, 100, 300)}      # NBB:135-136
    for k, p in staged.items():
        print("boosting, first %3d trees: rmse vs curve %.4f" % (k, np.sqrt(np.mean((p - y) ** 2))))
    forests = {}
    for n in (1, 3, 100):                                                               # NBB:204-212
        rf = lgb.train({"boosting": "rf", "objective": "regression", "num_leaves": 20, "min_data_in_leaf": 5,
                        "bagging_fraction": 0.632, "bagging_freq": 1, "feature_fraction": 1.0, "seed": 345,
                        "bagging_seed": 345, "verbose": -1}, dtrain, n)
        forests[n] = rf.predict(x)
        print("bagging, %3d trees:        rmse vs curve %.4f" % (n, np.sqrt(np.mean((forests[n] - y) ** 2))))
    if args.plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(1, 2, figsize=(12, 4))
        for a in ax:
            a.scatter(xtrain, ytrain, s=2, c="lightgrey")
            a.plot(x, y, c="black", lw=1)
        for k, p in staged.items():
            ax[0].plot(x, p, label=f"{k} trees")
        for n, p in forests.items():
            ax[1].plot(x, p, label=f"{n} trees")
        ax[0].set_title("boosting (staged prediction)"); ax[1].set_title("bagging / random forest")
        ax[0].legend(); ax[1].legend()
        fig.savefig(args.plot, dpi=120)


if __name__ == "__main__":
    main()
