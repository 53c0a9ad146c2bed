"""Rounds and materialised splits of every tree of the headline run (10.5 M x 28, 255 leaves, 500 trees).

    python tools/runs/rounds_per_tree.py [--precision full|q16] [--trees 500] [--rows 10500000] [--extra '{"k": v}']
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                   # noqa: E402
import lightgbm_amd as lgb                     # noqa: E402
from lightgbm_amd.utils.datasets import make_higgs_like   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="full")
ap.add_argument("--trees", type=int, default=500)
ap.add_argument("--rows", type=int, default=10_500_000)
ap.add_argument("--extra", default="")
args = ap.parse_args()
X, y = make_higgs_like(args.rows, seed=1234, device="cuda")
p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
     "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu", "use_quantized_grad": args.precision == "q16"}
if args.extra:
    p.update(json.loads(args.extra))
d = lgb.Dataset(X, label=y, params=p).construct()
b = lgb.Booster(p, d)
rounds, mat, ms = [], [], []
for i in range(args.trees):
    torch.cuda.synchronize()
    t = time.perf_counter()
    b.update()
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t) * 1e3)
    st = b.grow_stats()
    rounds.append(st["rounds"])
    mat.append(st["materialized_splits"])
r = torch.tensor(rounds, dtype=torch.float64)
m = torch.tensor(mat, dtype=torch.float64)
t = torch.tensor(ms, dtype=torch.float64)
print(f"trees {args.trees}: rounds mean {r.mean():.1f} median {r.median():.0f} max {int(r.max())} (tree {int(r.argmax())}); "
      f"materialised splits mean {m.mean():.0f} max {int(m.max())}; ms per tree mean {t.mean():.3f} median {t.median():.3f} max {t.max():.3f}")
for lo in range(0, args.trees, 50):
    hi = min(lo + 50, args.trees)
    print(f"  trees {lo:3d}-{hi - 1:3d}: rounds {r[lo:hi].mean():5.1f} (max {int(r[lo:hi].max()):3d})  splits {m[lo:hi].mean():5.0f}  ms {t[lo:hi].mean():.3f}")
