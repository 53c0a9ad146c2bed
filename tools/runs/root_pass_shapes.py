"""Root histogram pass against the data's shape (diagnosis of LDS-atomic rates): trains a few trees on
    higgs | randn28 | randn32 | higgs_cont (the Higgs-shaped table with its four 3-valued columns made continuous)
run it under rocprofv3 --kernel-trace and read grow_kernel<true, ...> from the kernel statistics."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                   # noqa: E402
import lightgbm_amd as lgb                     # noqa: E402
from lightgbm_amd.utils.datasets import make_higgs_like   # noqa: E402

kind = sys.argv[1]
n = 10_500_000
if kind.startswith("higgs"):
    X, y = make_higgs_like(n, seed=1234, device="cuda")
    if kind == "higgs_cont":
        for c in (8, 12, 16, 20):
            X[:, c] += 0.3 * torch.randn(n, device="cuda")
else:
    f = 32 if kind == "randn32" else 28
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(n, f, device="cuda", generator=g)
    y = (X[:, :4].sum(dim=1) + torch.randn(n, device="cuda", generator=g) > 0).float()
p = {"objective": "binary", "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
     "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "device_type": "gpu", "use_quantized_grad": False}
b = lgb.Booster(p, lgb.Dataset(X, label=y, params=p).construct())
for _ in range(12):
    b.update()
torch.cuda.synchronize()
