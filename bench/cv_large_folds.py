"""5-fold CV with folds far above the one-workgroup limit (1 M x 28 Higgs-shaped, 255 leaves): folds one after the other
against folds on concurrent host threads / HIP streams, and a single booster on 800 k rows for scale.

    python bench/cv_large_folds.py [--rows 1000000] [--rounds 40]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch                                   # noqa: E402
import lightgbm_amd as lgb                     # noqa: E402
from lightgbm_amd.utils.datasets import make_higgs_like   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--rounds", type=int, default=40)
args = ap.parse_args()
X, y = make_higgs_like(args.rows, seed=7, device="cuda")
p = {"objective": "binary", "metric": "binary_logloss", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1,
     "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu"}
d = lgb.Dataset(X, label=y, params=p).construct()
out = {}
for conc in (False, True):
    lgb.cv(p, d, 3, nfold=5, concurrent_folds=conc, stratified=False)
    torch.cuda.synchronize()
    t = time.time()
    r = lgb.cv(p, d, args.rounds, nfold=5, concurrent_folds=conc, stratified=False)
    torch.cuda.synchronize()
    dt = time.time() - t
    out[conc] = r["binary_logloss-mean"]
    print(f"{'concurrent' if conc else 'sequential'} folds: {args.rounds} rounds x 5 folds in {dt:.3f} s = {dt * 1000 / args.rounds:.2f} ms per round, "
          f"{dt * 1000 / args.rounds / 5:.2f} ms per fold-tree; logloss {r['binary_logloss-mean'][-1]:.6f}", flush=True)
print("identical numbers:", out[False] == out[True])
n = args.rows * 4 // 5
b = lgb.Booster(p, lgb.Dataset(X[:n], label=y[:n], params=p).construct())
for _ in range(3):
    b.update()
torch.cuda.synchronize()
t = time.time()
for _ in range(args.rounds):
    b.update()
torch.cuda.synchronize()
print(f"single booster on {n} rows: {(time.time() - t) * 1000 / args.rounds:.2f} ms per tree")
