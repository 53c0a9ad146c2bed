"""Quality check behind the benchmark: the same Higgs-shaped job on the HIP engine and on the CPU oracle.

    python bench/auc_parity.py [--rows 1000000 --trees 100]

Prints the AUC / logloss of both on a held-out part.  Binary gradients use exp(), which differs in the last
bits between libm and the device, so the models are not bit-identical (regression models are); the
metrics must agree to well below 1e-3.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                    # noqa: E402
import lightgbm_amd as lgb                                            # noqa: E402
from lightgbm_amd.utils.datasets import make_higgs_like               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--trees", type=int, default=100)
    args = ap.parse_args()
    X, y = make_higgs_like(args.rows + 200_000, seed=5, device="numpy")
    Xtr, ytr, Xte, yte = X[: args.rows], y[: args.rows], X[args.rows:], y[args.rows:]
    out = {}
    for dev in ("gpu", "cpu"):
        p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
             "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": dev, "metric": ["auc", "binary_logloss"]}
        dtr = lgb.Dataset(Xtr, label=ytr, params=p)
        dte = dtr.create_valid(Xte, label=yte)
        ev = {}
        t = time.time()
        lgb.train(p, dtr, args.trees, valid_sets=[dte], valid_names=["test"], evals_result=ev)
        out[dev] = (ev["test"]["auc"][-1], ev["test"]["binary_logloss"][-1], time.time() - t)
        print(f"{dev}: test auc {out[dev][0]:.6f}  logloss {out[dev][1]:.6f}  ({out[dev][2]:.1f} s incl. per-iteration eval)", flush=True)
    print(f"|auc difference| = {abs(out['gpu'][0] - out['cpu'][0]):.2e}")


if __name__ == "__main__":
    main()
