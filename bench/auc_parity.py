"""Quality check behind the benchmark: the Higgs-shaped job on the HIP engine (16-bit fixed-point gradient
histograms, leaf values renewed from 36-bit sums) against the CPU engine in BOTH precisions - the same
quantised arithmetic (the device's oracle) and full precision (36-bit gradient histograms, finer than the
float32 gradients): the reference the +-0.001 AUC bar of the benchmark is measured against.

    python bench/auc_parity.py [--rows 1000000 --trees 200]

Prints the held-out AUC / logloss of the three learners.  Binary gradients use exp(), which differs in the
last bits between libm and the device, so the GPU and CPU-quantised models are not bit-identical (regression
models are); their metrics agree to ~1e-6.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                    # noqa: E402,F401
import lightgbm_amd as lgb                                            # noqa: E402
from lightgbm_amd.utils.datasets import make_higgs_like               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--test-rows", type=int, default=500_000)
    ap.add_argument("--skip-cpu-quantised", action="store_true", help="leave out the CPU run in the device's own arithmetic")
    ap.add_argument("--log-every", type=int, default=0, help="print the evaluation every so many iterations (long runs)")
    ap.add_argument("--min-sum-hessian", type=float, default=100.0, help="the benchmark's 100; nearby values give the run-to-run band")
    ap.add_argument("--only", default="", help="run just this learner (gpu quantised / cpu quantised / cpu full precision)")
    args = ap.parse_args()
    X, y = make_higgs_like(args.rows + args.test_rows, seed=5, device="numpy")
    Xtr, ytr, Xte, yte = X[: args.rows], y[: args.rows], X[args.rows:], y[args.rows:]
    out = {}
    runs = [("gpu quantised", "gpu", True), ("cpu quantised", "cpu", True), ("cpu full precision", "cpu", False)]
    if args.skip_cpu_quantised:
        runs = [r for r in runs if r[0] != "cpu quantised"]
    if lgb.default_device() != "gpu":
        runs = [r for r in runs if r[1] != "gpu"]
    if args.only:
        runs = [r for r in runs if r[0] == args.only]
    for name, dev, quant in runs:
        p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
             "min_sum_hessian_in_leaf": args.min_sum_hessian, "verbose": -1, "device_type": dev, "metric": ["auc", "binary_logloss"],
             "use_quantized_grad": quant}
        dtr = lgb.Dataset(Xtr, label=ytr, params=p)
        dte = dtr.create_valid(Xte, label=yte)
        ev = {}
        t = time.time()
        lgb.train(p, dtr, args.trees, valid_sets=[dte], valid_names=["test"], evals_result=ev,
                  callbacks=[lgb.log_evaluation(args.log_every)])
        out[name] = {"auc": ev["test"]["auc"][-1], "logloss": ev["test"]["binary_logloss"][-1], "seconds": time.time() - t}
        print(f"{name:20s} test auc {out[name]['auc']:.6f}  logloss {out[name]['logloss']:.6f}  "
              f"({out[name]['seconds']:.1f} s incl. per-iteration eval)", flush=True)
    ref = out.get("cpu full precision", {}).get("auc")
    for name in out if ref is not None else []:
        if name != "cpu full precision":
            d = abs(out[name]["auc"] - ref)
            print(f"|auc({name}) - auc(cpu full precision)| = {d:.2e}  ({'within' if d <= 1e-3 else 'OUTSIDE'} the 0.001 bar)")
    print(json.dumps({"rows": args.rows, "trees": args.trees, "test_rows": args.test_rows,
                      "min_sum_hessian_in_leaf": args.min_sum_hessian, "results": out}))


if __name__ == "__main__":
    main()
