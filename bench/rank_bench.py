"""Learning-to-rank timing on an MSLR-WEB30K-shaped synthetic set (queries of ~120 documents, 136 features,
grades 0-4): seconds per boosting iteration with the per-query gradients taken on the device and - for
comparison - on the host (LGA_RANK_ON_HOST=1); likewise one evaluation of ndcg@10 over the training queries.

    python bench/rank_bench.py [rows] [trees]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightgbm_amd as lgb  # noqa: E402


def main():
    n_target = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    trees = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.RandomState(0)
    sizes = rng.randint(40, 200, size=n_target // 120)
    n = int(sizes.sum())
    X = rng.randn(n, 136).astype(np.float32)
    y = np.clip(np.round(X[:, 0] + np.sin(X[:, 1]) + 0.5 * X[:, 2] * X[:, 3] + 0.6 * rng.randn(n) + 1.0), 0, 4)
    p = {"objective": "lambdarank", "num_leaves": 255, "learning_rate": 0.1, "min_data_in_leaf": 50, "verbose": -1,
         "device_type": "gpu", "eval_at": [10]}
    out = {}
    for mode in ("device", "host"):
        if mode == "host":
            os.environ["LGA_RANK_ON_HOST"] = "1"
        d = lgb.Dataset(X, label=y, group=sizes, params=p).construct()
        b = lgb.Booster(p, d)
        for _ in range(3):
            b.update()
        import torch
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(trees):
            b.update()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / trees
        b._train_data_name = "training"
        ndcg = b.eval_train()[0][2]
        t0 = time.time()
        for _ in range(10):
            b.eval_train()
        te = (time.time() - t0) / 10
        out[mode] = dt
        print(f"{mode:6s} gradients: {n} rows, {len(sizes)} queries: {dt * 1e3:.2f} ms per iteration, train ndcg@10 {ndcg:.4f}; "
              f"one ndcg@10 evaluation ({mode}) {te * 1e3:.2f} ms", flush=True)
    print(f"speed-up of the device gradients: {out['host'] / out['device']:.1f}x")


if __name__ == "__main__":
    main()
