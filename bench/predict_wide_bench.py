"""Throughput of the device predictor on a wide matrix (the feature values stay in global memory): 200 columns.

    python bench/predict_wide_bench.py [--rows 1000000] [--cols 200] [--trees 300] [--leaves 255]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=200)
    ap.add_argument("--trees", type=int, default=300)
    ap.add_argument("--leaves", type=int, default=255)
    args = ap.parse_args()
    import numpy as np
    import torch
    import lightgbm_amd as lgb
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn(400_000, args.cols, device="cuda", generator=g)
    y = (X[:, ::7].sum(dim=1) + X[:, 1] * X[:, 2] + torch.randn(400_000, device="cuda", generator=g) > 0).float()
    p = {"objective": "binary", "num_leaves": args.leaves, "min_sum_hessian_in_leaf": 5.0, "verbose": -1, "device_type": "gpu"}
    b = lgb.train(p, lgb.Dataset(X, label=y, params=p), args.trees)
    Xp = torch.randn(args.rows, args.cols, device="cuda", generator=g)
    out = b.predict(Xp[:50000], device="gpu")
    torch.cuda.synchronize()
    t0 = time.time()
    out = b.predict(Xp, device="gpu")
    torch.cuda.synchronize()
    dt = time.time() - t0
    ref = b.predict(Xp[:20000].cpu().numpy(), device="cpu")
    print(json.dumps({"rows": args.rows, "cols": args.cols, "trees": b.num_trees(), "leaves": args.leaves, "seconds": round(dt, 4),
                      "rows_per_second": round(args.rows / dt), "max_abs_diff_vs_host": float(np.abs(ref - out[:20000]).max())}))


if __name__ == "__main__":
    main()
