"""Training with a validation set and a metric every iteration on large data: what evaluation adds per iteration."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_higgs_like
X, y = make_higgs_like(2_500_000, seed=3, device="cuda")
Xt, yt, Xv, yv = X[:2_000_000], y[:2_000_000], X[2_000_000:], y[2_000_000:]
for metric in (None, "binary_logloss", "auc"):
    p = {"objective": "binary", "num_leaves": 255, "learning_rate": 0.1, "min_sum_hessian_in_leaf": 100.0, "verbose": -1,
         "device_type": "gpu"}
    if metric:
        p["metric"] = metric
    d = lgb.Dataset(Xt, label=yt, params=p)
    v = d.create_valid(Xv, label=yv) if metric else None
    d.construct()
    torch.cuda.synchronize()
    t = time.time()
    ev = {}
    b = lgb.train(p, d, 100, valid_sets=[v] if metric else None, evals_result=ev, callbacks=[lgb.log_evaluation(0)])
    torch.cuda.synchronize()
    dt = time.time() - t
    last = ev["valid_0"][metric][-1] if metric else float("nan")
    print(f"metric {metric}: 100 iterations {dt:.2f} s ({dt * 10:.2f} ms per iteration), last value {last:.5f}", flush=True)
