"""Throughput of small boosters on one GPU: the 36.7 k x 6 training fold of the reference's grid search
(r/gridsearchCV.R:26-27,108-115), 127 leaves, grown (a) by the whole-tree kernel and (b) in batched rounds,
with 1 .. N boosters running concurrently, each driven by its own host thread on its own stream.

    python bench/small_tree_bench.py [--threads 1,4,8,16] [--iters 150] [--leaves 127]

Prints one JSON line per configuration: trees per second over all boosters and the time of one tree per booster.
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                   # noqa: E402
import torch                                                         # noqa: E402
import lightgbm_amd as lgb                                           # noqa: E402
from lightgbm_amd import _device, _lib                               # noqa: E402
from lightgbm_amd.utils.datasets import make_diamonds_like           # noqa: E402


def run(n_threads, iters, params, X, y):
    d = lgb.Dataset(X, label=y, params=params, free_raw_data=False).construct()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(n_threads)]
    boosters = []
    for s in streams:
        with torch.cuda.stream(s):
            boosters.append(lgb.Booster(params, d))
    torch.cuda.synchronize()
    barrier = threading.Barrier(n_threads + 1)

    def work(i):
        with torch.cuda.stream(streams[i]):
            _device.sync_stream(_lib.load()[0])
            for _ in range(5):
                boosters[i].update()
            barrier.wait()
            for _ in range(iters):
                boosters[i].update()
            streams[i].synchronize()
            barrier.wait()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.time()
    barrier.wait()
    dt = time.time() - t0
    for t in ts:
        t.join()
    return dt, boosters[0].grow_stats()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--iters", type=int, default=150)
    ap.add_argument("--leaves", type=int, default=127)
    ap.add_argument("--rows", type=int, default=36700)
    args = ap.parse_args()
    X, y = make_diamonds_like(args.rows, seed=0)
    for whole in (65536, 0):
        for nt in [int(v) for v in args.threads.split(",")]:
            p = {"objective": "regression", "num_leaves": args.leaves, "learning_rate": 0.05, "min_data_in_leaf": 20,
                 "bagging_fraction": 0.8, "bagging_freq": 4, "feature_fraction": 0.8, "verbose": -1,
                 "whole_tree_max_rows": whole}
            dt, st = run(nt, args.iters, p, X, y)
            print(json.dumps({"mode": "whole-tree kernel" if whole else "batched rounds", "boosters": nt, "iters": args.iters,
                              "rows": args.rows, "leaves": args.leaves, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                              "trees_per_s": round(nt * args.iters / dt, 1), "ms_per_tree_per_booster": round(1e3 * dt / args.iters, 4),
                              "rounds_last_tree": st["rounds"]}), flush=True)


if __name__ == "__main__":
    main()
