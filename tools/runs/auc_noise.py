"""How much does the held-out AUC of the headline job move under perturbations that should not matter
(min_sum_hessian_in_leaf 99 ... 101)?  The yardstick for reading bench/auc_parity.py at the full row count."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_higgs_like
rows, test_rows, trees = 10_500_000, 500_000, int(sys.argv[1]) if len(sys.argv) > 1 else 200
X, y = make_higgs_like(rows + test_rows, seed=5, device="numpy")
Xtr, ytr, Xte, yte = X[:rows], y[:rows], X[rows:], y[rows:]
res = {}
for msh in [100.0, 99.0, 101.0, 100.5, 98.0, 102.0]:
    p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
         "min_sum_hessian_in_leaf": msh, "verbose": -1, "device_type": "gpu", "metric": ["auc"]}
    dtr = lgb.Dataset(Xtr, label=ytr, params=p)
    dte = dtr.create_valid(Xte, label=yte)
    ev = {}
    lgb.train(p, dtr, trees, valid_sets=[dte], valid_names=["test"], evals_result=ev, callbacks=[lgb.log_evaluation(0)])
    res[msh] = [ev["test"]["auc"][i] for i in range(49, trees, 50)]
    print("min_sum_hessian_in_leaf", msh, "test auc at 50, 100, ...:", " ".join("%.6f" % v for v in res[msh]), flush=True)
last = [v[-1] for v in res.values()]
print("spread of the final AUC: %.2e" % (max(last) - min(last)))
