"""Per-tree time on the shapes of upstream's GPU benchmark list (synthetic data of those shapes) at its settings:
255 leaves, 255 bins, lr 0.1, min_data_in_leaf 1, min_sum_hessian_in_leaf 100.

    python tools/runs/wide_shapes.py [shape prefix] [--precision full|q16] [--trees 30]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                   # noqa: E402
import lightgbm_amd as lgb                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="?", default="")
ap.add_argument("--precision", default="full", choices=["full", "q16"])
ap.add_argument("--trees", type=int, default=30)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()
shapes = [("MS-LTR-shaped", 2_270_000, 137, "regression"), ("Yahoo-LTR-shaped", 473_000, 700, "regression"),
          ("Bosch-shaped", 1_000_000, 968, "binary"), ("Epsilon-shaped", 400_000, 2000, "binary"),
          ("Expo-shaped", 11_000_000, 700, "binary")]
if args.shape:   # one shape only (for a profiler run): wide_shapes.py Bosch
    shapes = [s for s in shapes if s[0].startswith(args.shape)]
for name, n, f, objective in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.empty(n, f, device="cuda")
    for lo in range(0, n, 1 << 20):   # in row blocks: randn over the whole 11 M x 700 matrix needs a second copy of it
        X[lo:lo + (1 << 20)].normal_(generator=g)
    s = X[:, :8].sum(dim=1) + 0.5 * X[:, 8] * X[:, 9] + torch.randn(n, device="cuda", generator=g)
    y = (s > 0).float() if objective == "binary" else s.float()
    p = {"objective": objective, "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
         "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu", "use_quantized_grad": args.precision == "q16"}
    t = time.time()
    d = lgb.Dataset(X, label=y, params=p).construct()
    torch.cuda.synchronize()
    t_bin = time.time() - t
    del X
    b = lgb.Booster(p, d)
    for _ in range(args.warmup):
        b.update()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(args.trees):
        b.update()
    torch.cuda.synchronize()
    dt = time.time() - t
    st = b.grow_stats()
    print(f"{name} {n} x {f} ({objective}, {args.precision}): binning {t_bin:.2f} s, {dt * 1000 / args.trees:.2f} ms per tree; "
          f"last tree: {st}", flush=True)
    del b, d
    torch.cuda.empty_cache()
