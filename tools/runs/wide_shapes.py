"""Per-tree time on shapes of upstream's GPU benchmark list (synthetic data of those shapes; 255 leaves, 255 bins)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import lightgbm_amd as lgb
shapes = [("MS-LTR-shaped", 2_270_000, 137), ("Yahoo-LTR-shaped", 473_000, 700), ("Bosch-shaped", 1_000_000, 968)]
if len(sys.argv) > 1:   # one shape only (for a profiler run): wide_shapes.py Bosch
    shapes = [s for s in shapes if s[0].startswith(sys.argv[1])]
for name, n, f in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(n, f, device="cuda", generator=g)
    y = (X[:, :8].sum(dim=1) + 0.5 * X[:, 8] * X[:, 9] + torch.randn(n, device="cuda", generator=g)).float()
    p = {"objective": "regression", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 20,
         "verbose": -1, "device_type": "gpu"}
    t = time.time()
    d = lgb.Dataset(X, label=y, params=p).construct()
    torch.cuda.synchronize()
    t_bin = time.time() - t
    del X
    b = lgb.Booster(p, d)
    for _ in range(3):
        b.update()
    torch.cuda.synchronize()
    t = time.time()
    k = 30
    for _ in range(k):
        b.update()
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"{name} {n} x {f}: binning {t_bin:.2f} s, {dt * 1000 / k:.2f} ms per tree", flush=True)
    del b, d
    torch.cuda.empty_cache()
