import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_higgs_like
p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "verbose": -1, "device_type": "gpu", "use_quantized_grad": False,
     "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 100.0}
for rep in range(3):
    X, y = make_higgs_like(10_500_000, seed=1234 + rep, device="cuda")
    torch.cuda.synchronize()
    t = time.time()
    ds = lgb.Dataset(X, label=y, params=p).construct()
    torch.cuda.synchronize()
    t1 = time.time()
    b = lgb.Booster(p, ds)
    torch.cuda.synchronize()
    t2 = time.time()
    b.update()
    torch.cuda.synchronize()
    t3 = time.time()
    print(f"rep {rep}: construct {t1 - t:.3f} s, booster {t2 - t1:.3f} s, first tree {t3 - t2:.3f} s", flush=True)
    del b, ds, X, y
import cProfile, pstats
X, y = make_higgs_like(10_500_000, seed=99, device="cuda")
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
ds = lgb.Dataset(X, label=y, params=p).construct(); torch.cuda.synchronize()
b = lgb.Booster(p, ds); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
