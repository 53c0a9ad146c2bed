"""Wide dense data (Epsilon-shaped: 400 k x 2000): feature-parallel training, and a 64-point grid whose
points are dealt out to the GPUs.

    python examples/epsilon_feature_parallel.py [--rows 400000 --features 2000 --trees 20]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/epsilon_feature_parallel.py --grid 64

Feature-parallel: every rank holds all rows, builds histograms and scans splits only for its own
32-feature tiles, and the ranks exchange their 32-byte winners per leaf (no histogram traffic).
Grid mode (--grid K): rank r evaluates grid points r, r + world, ... with a 3-fold CV each and rank 0
prints the ranking (task parallelism; no collective inside a model).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import lightgbm_amd as lgb                                                       # noqa: E402
from lightgbm_amd.utils.datasets import make_epsilon_like                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=400_000)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--trees", type=int, default=20)
    ap.add_argument("--leaves", type=int, default=255)
    ap.add_argument("--grid", type=int, default=0, help="evaluate this many grid points instead of one model")
    ap.add_argument("--device", default=lgb.default_device())
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.device == "gpu":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if args.device == "gpu" else "gloo")
    X, y = make_epsilon_like(args.rows, args.features, seed=7)       # the same data on every rank
    # the settings of upstream's GPU benchmark: 255 leaves, 255 bins, min_data_in_leaf 1, min_sum_hessian_in_leaf 100
    base = {"objective": "binary", "num_leaves": args.leaves, "min_data_in_leaf": 1, "min_sum_hessian_in_leaf": 100.0,
            "max_bin": 255, "learning_rate": 0.1, "verbose": -1, "device_type": args.device}
    if args.grid <= 0:
        p = {**base, "tree_learner": "feature" if world > 1 else "serial", "metric": "auc"}
        from lightgbm_amd.parallel import default_comm
        comm = default_comm() if world > 1 else None
        t = time.time()
        ds = lgb.Dataset(X, label=y, params=p).construct()
        t_bin = time.time() - t
        t = time.time()
        ev = {}
        b = lgb.train(p, ds, args.trees, comm=comm, valid_sets=[ds], evals_result=ev)
        dt = time.time() - t
        if rank == 0:
            print(f"feature-parallel over {world} rank(s): binning {t_bin:.2f} s, {args.trees} trees {dt:.2f} s "
                  f"({dt * 1000 / args.trees:.1f} ms/tree), train auc {ev['training']['auc'][-1]:.4f}", flush=True)
        if comm is not None:
            comm.close()
    else:
        from lightgbm_amd.grid import expand_grid, grid_search_cv, grid_search_cv_distributed, rank_grid
        grid = expand_grid(iteration=None, score=None, learning_rate=[0.2, 0.1], num_leaves=[31, 63, 127, 255],
                           min_data_in_leaf=[20, 100], lambda_l2=[0.0, 1.0], feature_fraction=[0.5, 1.0])
        grid = grid[: args.grid]
        t = time.time()
        shared = {k: v for k, v in base.items() if k not in ("objective", "num_leaves", "min_data_in_leaf", "learning_rate")}
        search = grid_search_cv_distributed if world > 1 else grid_search_cv
        done = search(grid, X=X, y=y, objective="binary", eval="binary_logloss", nrounds=args.trees, nfold=3,
                      early_stopping_rounds=5, base_params=shared, dataset_params=shared)
        if rank == 0:
            print(f"{len(done)} grid points on {world} rank(s): {time.time() - t:.1f} s")
            print(rank_grid(done)[:5])
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
