import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like
X, y = make_diamonds_like(53940, seed=0)
p = {"objective": "regression", "num_leaves": 127, "learning_rate": 0.01, "verbose": -1, "device_type": "gpu", "metric": "rmse",
     "feature_fraction": 0.8, "bagging_fraction": 0.8, "bagging_freq": 4}
d = lgb.Dataset(X[:45849], label=y[:45849], params=p).construct()
idx = np.arange(45849)
tr = d.subset(idx[idx % 5 != 0]).construct(); te = d.subset(idx[idx % 5 == 0]).construct()
bs = []
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(100):
        b = lgb.Booster(p, tr); b.add_valid(te, "valid"); bs.append(b)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"create + add_valid: {dt * 10:.3f} ms per booster", flush=True)
    t = time.time(); bs.clear(); torch.cuda.synchronize(); print(f"free: {(time.time() - t) * 10:.3f} ms per booster", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    b = lgb.Booster(p, tr); b.add_valid(te, "valid"); bs.append(b)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
