"""What releasing the boosters of a grid costs: 48 rows x 5 folds alive at once (as in the reference grid),
advanced a few lock-step chunks, then released one by one with the time of every release.
    python tools/runs/grid_booster_free_time.py [iterations]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.rstyle import lgb_cv_steps, lgb_Dataset
from lightgbm_amd.utils.datasets import make_diamonds_like
from lightgbm_amd.basic import Booster

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
X, y = make_diamonds_like(53940, seed=0)
d = lgb_Dataset(X[:45849], label=y[:45849])
rows = []
for k in range(48):
    p = {"learning_rate": 0.01, "num_leaves": 127, "min_data_in_leaf": 20, "feature_fraction": 0.8,
         "bagging_fraction": 0.8, "bagging_freq": 4}
    g = lgb_cv_steps(p, d, nrounds=1000, nfold=5, objective="regression", eval="rmse", showsd=False,
                     early_stopping_rounds=5, verbose=0, rollback_overrun=False)
    rows.append([g, next(g)])
done = 0
while done < iters:
    boosters = [b for _, req in rows for b in req[0]]
    n = min(req[1] for _, req in rows)
    values, no_split, it = Booster.update_group(boosters, n, True)
    at = 0
    for r in rows:
        k = len(r[1][0])
        r[1] = r[0].send((values[:, at:at + k], no_split[:, at:at + k], it))
        at += k
    done += it
boosters = [b for _, req in rows for b in req[0]]
times = []
for b in boosters:
    t = time.perf_counter(); b._free(); times.append((time.perf_counter() - t) * 1e3)
t = np.array(times)
print(f"{len(t)} boosters of {done} iterations: release mean {t.mean():.3f} ms, median {np.median(t):.3f}, max {t.max():.3f}; "
      f"first 20 mean {t[:20].mean():.3f}, last 100 mean {t[-100:].mean():.3f}; total {t.sum():.0f} ms")
