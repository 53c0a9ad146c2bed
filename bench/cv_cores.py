"""5-fold CV wall time of the reference's first cv call (r/gridsearchCV.R:70-78: 45.8 k x 6, learning rate 0.1, early
stopping 5) with the fold boosters driven (a) by five host threads that wait for their growth rounds - the wait spins
briefly, then yields - and (b) in lockstep chunks on the device.  Run it with and without `taskset -c 0,1`: five
waiting threads on two cores must not cost much more than on sixteen.

    python bench/cv_cores.py            taskset -c 0,1 python bench/cv_cores.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                    # noqa: E402
import lightgbm_amd as lgb                                            # noqa: E402
from lightgbm_amd.utils.datasets import make_diamonds_like            # noqa: E402

X, y = make_diamonds_like(45840, seed=0)
p = {"objective": "regression", "learning_rate": 0.1, "metric": "rmse", "verbose": -1}
out = {"cpus": len(os.sched_getaffinity(0))}
for name, kw in (("fold threads", {"lockstep_chunk": 0, "concurrent_folds": True}), ("lockstep chunks", {})):
    best = 1e9
    for rep in range(4):
        d = lgb.Dataset(X, label=y, params=p)
        t = time.perf_counter()
        r = lgb.cv(p, d, 1000, nfold=5, early_stopping_rounds=5, seed=1, **kw)
        dt = time.perf_counter() - t
        if rep:
            best = min(best, dt)
    out[name] = {"seconds": round(best, 4), "iterations": len(r["rmse-mean"])}
print(json.dumps(out))
