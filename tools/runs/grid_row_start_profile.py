import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.rstyle import lgb_cv_steps, lgb_Dataset
from lightgbm_amd.utils.datasets import make_diamonds_like
X, y = make_diamonds_like(53940, seed=0)
d = lgb_Dataset(X[:45849], label=y[:45849])
def start(lr):
    p = {"learning_rate": lr, "num_leaves": 127, "min_data_in_leaf": 20, "feature_fraction": 0.8, "bagging_fraction": 0.8, "bagging_freq": 4}
    g = lgb_cv_steps(p, d, nrounds=1000, nfold=5, objective="regression", eval="rmse", showsd=False, early_stopping_rounds=5, verbose=0, rollback_overrun=False)
    return g, next(g)
keep = [start(0.1) for _ in range(3)]
t = time.time(); keep += [start(0.05) for _ in range(40)]; print("per row start:", (time.time() - t) / 40 * 1000, "ms")
pr = cProfile.Profile(); pr.enable(); keep += [start(0.02) for _ in range(40)]; pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
