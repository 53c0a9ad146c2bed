"""Bagging vs boosting at scale: rf mode (bagging_fraction 0.63, feature_fraction 0.8) against gbdt on a
synthetic 1 M x 50 regression problem.

    python examples/rf_vs_gbdt_1m.py [--rows 1000000 --features 50 --trees 100]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/rf_vs_gbdt_1m.py

With several ranks the rows are sharded and both models train data-parallel (one RCCL all-reduce of the
new leaf histograms per growth round); every rank ends with the same model.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import lightgbm_amd as lgb                                                       # noqa: E402
from lightgbm_amd.utils.datasets import make_regression                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--features", type=int, default=50)
    ap.add_argument("--trees", type=int, default=100)
    ap.add_argument("--leaves", type=int, default=127)
    ap.add_argument("--device", default=lgb.default_device())
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    comm = None
    if world > 1:
        import torch
        import torch.distributed as dist
        on_gpu = args.device == "gpu"
        if on_gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if on_gpu else "gloo")
    lo, hi = args.rows * rank // world, args.rows * (rank + 1) // world
    X, y = make_regression(hi - lo, args.features, seed=100 + rank, coef_seed=0)
    Xt, yt = make_regression(50_000, args.features, seed=99, coef_seed=0)
    base = {"objective": "regression", "num_leaves": args.leaves, "min_data_in_leaf": 20, "verbose": -1,
            "device_type": args.device, "tree_learner": "data" if world > 1 else "serial"}
    configs = {
        "rf": {**base, "boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1, "feature_fraction": 0.8},
        "gbdt": {**base, "boosting": "gbdt", "learning_rate": 0.1},
    }
    for name, p in configs.items():
        if world > 1:
            from lightgbm_amd.parallel import default_comm, distributed_dataset
            ds = distributed_dataset(X, y, params=p)
            comm = default_comm()
        else:
            ds = lgb.Dataset(X, label=y, params=p)
        lgb.train(p, ds, 2, comm=comm)          # warm-up: code objects, allocator pools
        t = time.time()
        b = lgb.train(p, ds, args.trees, comm=comm)
        dt = time.time() - t
        rmse = float(np.sqrt(np.mean((b.predict(Xt) - yt) ** 2)))
        if rank == 0:
            print(f"{name:5s} {args.trees} trees on {args.rows} x {args.features} ({world} rank(s), {args.device}): "
                  f"{dt:.2f} s, {dt * 1000 / args.trees:.2f} ms/tree, test rmse {rmse:.4f} (label sd {float(np.std(yt)):.4f})",
                  flush=True)
        if comm is not None:
            comm.close()
            comm = None
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
