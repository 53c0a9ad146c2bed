import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.rstyle import lgb_cv, lgb_Dataset
from lightgbm_amd.utils.datasets import make_diamonds_like
X, y = make_diamonds_like(53940, seed=0)
n_tr = 45849
d = lgb_Dataset(X[:n_tr], label=y[:n_tr])
for leaves in (31, 63, 127):
    p = {"learning_rate": 0.01, "num_leaves": leaves, "min_data_in_leaf": 20, "feature_fraction": 0.8, "bagging_fraction": 0.8, "bagging_freq": 4}
    for rep in range(2):
        t = time.time()
        cvm = lgb_cv(p, d, nrounds=300, nfold=5, objective="regression", eval="rmse", showsd=False, early_stopping_rounds=5, verbose=0)
        dt = time.time() - t
    print(f"leaves {leaves}: {dt:.3f} s for {cvm.best_iter} rounds x 5 folds -> {dt * 1000 / max(cvm.best_iter, 1):.2f} ms per round", flush=True)
