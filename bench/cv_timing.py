import time, sys
sys.path.insert(0, ".")
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like
X, y = make_diamonds_like(45000, seed=1)
for dev in ("gpu", "cpu"):
    d = lgb.Dataset(X, label=y, params={"device_type": dev, "verbose": -1})
    p = {"device_type": dev, "verbose": -1, "objective": "regression", "metric": "rmse", "learning_rate": 0.1, "num_leaves": 63, "min_data_in_leaf": 20}
    for conc in ((False, True) if dev == "gpu" else (False,)):
        lgb.cv(p, d, 5, nfold=5, concurrent_folds=conc)
        t = time.time()
        r = lgb.cv(p, d, 100, nfold=5, concurrent_folds=conc)
        dt = time.time() - t
        print(dev, "concurrent" if conc else "sequential", "100 rounds x 5 folds: %.3f s  (%.2f ms per tree)  rmse %.5f" % (dt, dt * 1000 / 500, r["rmse-mean"][-1]), flush=True)
