"""Throughput of the device ensemble predictor: Higgs-shaped rows through a 500-tree, 255-leaf binary model.

    python bench/predict_bench.py [--rows 4000000] [--trees 500] [--leaves 255]

The matrix lives in HBM (a float32 torch tensor), so the number is the kernel's, not the PCIe link's."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--train-rows", type=int, default=1_000_000)
    ap.add_argument("--trees", type=int, default=500)
    ap.add_argument("--leaves", type=int, default=255)
    args = ap.parse_args()
    import numpy as np
    import torch
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_higgs_like
    X, y = make_higgs_like(args.train_rows, seed=1, device="cuda")
    p = {"objective": "binary", "num_leaves": args.leaves, "min_sum_hessian_in_leaf": 10.0, "verbose": -1,
         "device_type": "gpu"}
    b = lgb.train(p, lgb.Dataset(X, label=y, params=p), args.trees)
    Xp, _ = make_higgs_like(args.rows, seed=2, device="cuda")
    Xp = Xp.float().contiguous()
    out = b.predict(Xp[:100000], device="gpu")           # warm-up (model packing, kernel load)
    torch.cuda.synchronize()
    t0 = time.time()
    out = b.predict(Xp, device="gpu")
    torch.cuda.synchronize()
    dt = time.time() - t0
    ref = b.predict(Xp[:20000].cpu().numpy(), device="cpu")
    err = float(np.abs(ref - out[:20000]).max())
    print(json.dumps({"rows": args.rows, "trees": b.num_trees(), "leaves": args.leaves, "seconds": round(dt, 4),
                      "rows_per_second": round(args.rows / dt), "ns_per_row_tree": round(dt * 1e9 / args.rows / b.num_trees(), 4),
                      "max_abs_diff_vs_host": err}))


if __name__ == "__main__":
    main()
