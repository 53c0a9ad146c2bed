"""Bagging vs boosting at scale: rf mode (bagging_fraction 0.63, feature_fraction 0.8) against gbdt on a
synthetic 1 M x 50 regression problem.

    python examples/rf_vs_gbdt_1m.py [--rows 1000000 --features 50 --trees 100]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/rf_vs_gbdt_1m.py

With several ranks each model trains the way that suits it.  The trees of a bagged forest do not depend on each
other ("fitted in parallel", bagging_boosting.ipynb:11): every rank holds all the rows, grows its share of the
trees on its own GPU and the forests are concatenated (``lightgbm_amd.parallel.train_forest_tree_parallel``; the
merged forest equals the one-process forest tree for tree).  Boosted trees depend on each other: gbdt shards the
rows and trains data-parallel (one RCCL exchange of the new leaf histograms per growth round).  Every rank ends
with the same two models.
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
    # the table is made of `world` blocks; block r is rank r's shard when the rows are sharded
    bounds = [args.rows * r // world for r in range(world + 1)]
    blocks = [make_regression(bounds[r + 1] - bounds[r], args.features, seed=100 + r, coef_seed=0) for r in range(world)]
    X, y = blocks[rank]
    Xt, yt = make_regression(50_000, args.features, seed=99, coef_seed=0)
    base = {"objective": "regression", "num_leaves": args.leaves, "min_data_in_leaf": 20, "verbose": -1,
            "device_type": args.device}
    configs = {
        "rf": {**base, "boosting": "rf", "bagging_fraction": 0.63, "bagging_freq": 1, "feature_fraction": 0.8},
        "gbdt": {**base, "boosting": "gbdt", "learning_rate": 0.1, "tree_learner": "data" if world > 1 else "serial"},
    }
    for name, p in configs.items():
        how = "one process"
        if world > 1 and name == "rf":
            from lightgbm_amd.parallel import train_forest_tree_parallel
            Xa, ya = np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks])
            train_forest_tree_parallel(p, Xa, ya, 2 * world)      # warm-up: code objects, allocator pools
            dist.barrier()
            t = time.time()
            b = train_forest_tree_parallel(p, Xa, ya, args.trees)
            dt = time.time() - t
            how = f"tree-parallel, {world} ranks x {args.trees // world} trees on all rows"
            del Xa, ya
        else:
            if world > 1:
                from lightgbm_amd.parallel import default_comm, distributed_dataset
                ds = distributed_dataset(X, y, params=p)
                comm = default_comm()
                how = f"data-parallel, {world} ranks x {len(y)} rows"
            else:
                ds = lgb.Dataset(X, label=y, params=p)
            lgb.train(p, ds, 2, comm=comm)          # warm-up
            t = time.time()
            b = lgb.train(p, ds, args.trees, comm=comm)
            dt = time.time() - t
        rmse = float(np.sqrt(np.mean((b.predict(Xt) - yt) ** 2)))
        if rank == 0:
            print(f"{name:5s} {args.trees} trees on {args.rows} x {args.features} ({how}, {args.device}): "
                  f"{dt:.2f} s, {dt * 1000 / args.trees:.2f} ms/tree, test rmse {rmse:.4f} (label sd {float(np.std(yt)):.4f})",
                  flush=True)
        if comm is not None:
            comm.close()
            comm = None
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
