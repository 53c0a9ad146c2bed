"""Learning to rank on a synthetic search log: queries of 20-120 documents with graded relevance (0-4).

lambdarank and rank_xendcg through the training API (query sizes go into ``Dataset(group=)``), early stopping
on ndcg@10 of held-out queries, query-wise cross-validation, and the scikit-learn style ``LGBMRanker``.

    python examples/learning_to_rank.py [--device cpu|gpu] [--queries 3000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightgbm_amd as lgb  # noqa: E402
from lightgbm_amd.sklearn import LGBMRanker  # noqa: E402


def make_queries(n_queries, n_features=30, seed=0):
    rng = np.random.RandomState(seed)
    sizes = rng.randint(20, 120, size=n_queries)
    n = int(sizes.sum())
    X = rng.randn(n, n_features).astype(np.float32)
    topic = np.repeat(rng.randn(n_queries), sizes)                 # a per-query offset no ranker should care about
    rel = X[:, 0] + 0.8 * np.tanh(X[:, 1] * X[:, 2]) + 0.5 * X[:, 3] + 0.7 * rng.randn(n)
    y = np.clip(np.round(rel + 1.2), 0, 4)
    X[:, 4] += topic
    return X, y, sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default=lgb.default_device())
    ap.add_argument("--queries", type=int, default=3000)
    args = ap.parse_args()
    X, y, sizes = make_queries(args.queries)
    n_train_q = int(0.8 * args.queries)
    cut = int(sizes[:n_train_q].sum())
    base = {"device_type": args.device, "verbose": -1, "num_leaves": 63, "min_data_in_leaf": 50, "learning_rate": 0.1,
            "eval_at": [10]}
    dtrain = lgb.Dataset(X[:cut], label=y[:cut], group=sizes[:n_train_q], params=base)
    dvalid = lgb.Dataset(X[cut:], label=y[cut:], group=sizes[n_train_q:], reference=dtrain)
    for objective in ("lambdarank", "rank_xendcg"):
        t0 = time.time()
        bst = lgb.train({**base, "objective": objective}, dtrain, 300, valid_sets=[dvalid], valid_names=["held-out"],
                        early_stopping_rounds=20, verbose_eval=False)
        print(f"{objective:12s} best iteration {bst.best_iteration:3d}  held-out "
              + "  ".join(f"{k} {v:.4f}" for k, v in bst.best_score["held-out"].items()) + f"   ({time.time() - t0:.1f} s)")
    cv = lgb.cv({**base, "objective": "lambdarank"}, lgb.Dataset(X, label=y, group=sizes, params=base), 50, nfold=4)
    print(f"4-fold query-wise CV, 50 rounds: ndcg@10 {cv['ndcg@10-mean'][-1]:.4f} +- {cv['ndcg@10-stdv'][-1]:.4f}")
    ranker = LGBMRanker(n_estimators=100, num_leaves=63, min_child_samples=50, device_type=args.device)
    ranker.fit(X[:cut], y[:cut], group=sizes[:n_train_q], eval_set=[(X[cut:], y[cut:])], eval_group=[sizes[n_train_q:]], eval_at=[10])
    scores = ranker.predict(X[cut:cut + sizes[n_train_q]])
    print("LGBMRanker ndcg@10", round(ranker.evals_result_["valid_0"]["ndcg@10"][-1], 4),
          "| first held-out query, documents by predicted score:", np.argsort(-scores)[:5].tolist())


if __name__ == "__main__":
    main()
