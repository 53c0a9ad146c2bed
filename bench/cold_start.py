"""Where does the first model's wall time go?  The reference's first fit - lgb.train, 200 rounds, learning rate 0.1
on the 45.8 k x 6 training table (r/gridsearchCV.R:55-61; 1.02 s on its author's machine, "LightGBM R.ipynb" raw
line 318) - timed in a FRESH process, phase by phase, and once more warm.

    python bench/cold_start.py
"""
import json
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                    # noqa: E402

t_np = time.perf_counter()
import lightgbm_amd as lgb                                            # noqa: E402
from lightgbm_amd import _lib                                         # noqa: E402
from lightgbm_amd.rstyle import lgb_Dataset, lgb_train, predict       # noqa: E402
from lightgbm_amd.utils.datasets import make_diamonds_like            # noqa: E402

t_import = time.perf_counter()
_lib.load()
t_lib = time.perf_counter()
X, y = make_diamonds_like(53940, seed=0)
rng = np.random.RandomState(3928272)
inn = rng.choice([False, True], size=len(y), replace=True, p=[0.15, 0.85])
Xtr, ytr = X[inn], y[inn]
t_data = time.perf_counter()


def fit():
    t = [time.perf_counter()]
    d = lgb_Dataset(Xtr, label=ytr)
    d.construct()
    t.append(time.perf_counter())
    b = lgb.Booster({"learning_rate": 0.1, "objective": "regression", "verbose": -1}, d)
    t.append(time.perf_counter())
    b.update()
    t.append(time.perf_counter())
    d2 = lgb_Dataset(Xtr, label=ytr)
    t.append(time.perf_counter())
    m = lgb_train(params={"learning_rate": 0.1}, data=d2, nrounds=200, objective="regression", verbose=0)
    t.append(time.perf_counter())
    p = predict(m, Xtr[:1000])
    t.append(time.perf_counter())
    return {"dataset_construct": t[1] - t[0], "booster_create": t[2] - t[1], "first_iteration": t[3] - t[2],
            "lgb_train_200_rounds_incl_dataset": t[5] - t[4], "predict_1000_rows": t[6] - t[5]}


cold = fit()
warm = fit()
out = {"device": lgb.default_device(), "rows": int(inn.sum()),
       "process": {"numpy_import": t_np - T0, "package_import_incl_torch": t_import - t_np, "native_library_load": t_lib - t_import,
                   "synthetic_data": t_data - t_lib},
       "cold": cold, "warm": warm,
       "first_model_wall_seconds": cold["dataset_construct"] + cold["booster_create"] + cold["first_iteration"]
       + cold["lgb_train_200_rounds_incl_dataset"]}
print(json.dumps(out, indent=1))
