"""Chunked boosting on the device against the step-by-step paths: same trees, same metric values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like, make_regression

def trees_of(b):
    s = b.model_to_string()
    return s[s.index("Tree=0"):s.index("end of trees")]

def one(name, X, y, p, rounds=40, nfold=4):
    out = {}
    for tag, dev, chunk in (("cpu", "cpu", 0), ("gpu plain", "gpu", 0), ("gpu chunk", "gpu", 7)):
        pp = {**p, "device_type": dev, "verbose": -1}
        t = time.time()
        r = lgb.cv(pp, lgb.Dataset(X, label=y, params=pp), rounds, nfold=nfold, early_stopping_rounds=5, seed=3,
                   lockstep_chunk=chunk, concurrent_folds=False, return_cvbooster=True)
        out[tag] = (r, time.time() - t)
    ks = [k for k in out["cpu"][0] if k != "cvbooster"]
    ok = True
    for tag in ("gpu plain", "gpu chunk"):
        same_trees = all(trees_of(a) == trees_of(b) for a, b in zip(out["cpu"][0]["cvbooster"].boosters, out[tag][0]["cvbooster"].boosters))
        same_len = all(len(out[tag][0][k]) == len(out["cpu"][0][k]) for k in ks)
        bit = same_len and all(out[tag][0][k] == out["gpu plain"][0][k] for k in ks)
        close = same_len and all(np.allclose(out[tag][0][k], out["cpu"][0][k], rtol=1e-9) for k in ks)
        print(f"{name:28s} {tag:10s} trees==cpu {same_trees}  metrics==gpu-plain(bitwise) {bit}  ~cpu {close}  "
              f"iters {len(out[tag][0][ks[0]])}  {out[tag][1]:.2f}s", flush=True)
        ok = ok and same_trees and bit and close
    return ok

ok = True
X, y = make_diamonds_like(45000, seed=0)
ok &= one("diamonds bag+ff", X, y, {"objective": "regression", "num_leaves": 63, "learning_rate": 0.1, "min_data_in_leaf": 20,
                                   "bagging_fraction": 0.8, "bagging_freq": 4, "feature_fraction": 0.8})
ok &= one("diamonds 127 leaves", X, y, {"objective": "regression", "num_leaves": 127, "learning_rate": 0.3, "min_data_in_leaf": 40, "metric": ["l2", "l1"]})
X2, y2 = make_regression(20000, 12, seed=2)
X2[::9, 3] = np.nan
w = None
ok &= one("huber 12 features nan", X2, y2, {"objective": "huber", "num_leaves": 31, "learning_rate": 0.2, "metric": "rmse", "max_depth": 6,
                                           "bagging_fraction": 0.5, "bagging_freq": 1, "extra_trees": True})
X3, y3 = make_regression(30000, 28, seed=5)
ok &= one("28 features bynode", X3, (y3 > 0).astype(float), {"objective": "binary", "num_leaves": 15, "learning_rate": 0.2,
                                                             "metric": ["binary_logloss", "binary_error"], "feature_fraction_bynode": 0.6})
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
