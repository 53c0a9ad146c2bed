import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_higgs_like
X, y = make_higgs_like(1_000_000, seed=1, device="cuda")
p = {"objective": "binary", "num_leaves": 255, "min_sum_hessian_in_leaf": 10.0, "verbose": -1, "device_type": "gpu"}
b = lgb.train(p, lgb.Dataset(X, label=y, params=p), 500)
Xp, _ = make_higgs_like(4_000_000, seed=2, device="cuda")
Xh = Xp.float().cpu().numpy()
out = b.predict(Xh[:100000], device="gpu")
for rep in range(3):
    t0 = time.time(); out = b.predict(Xh, device="gpu"); dt = time.time() - t0
    print(json.dumps({"host_matrix_rows_per_second": round(len(Xh) / dt), "seconds": round(dt, 4)}))
Xd = Xp.float().contiguous(); torch.cuda.synchronize()
t0 = time.time(); o2 = b.predict(Xd, device="gpu"); torch.cuda.synchronize(); print("device matrix", round(time.time() - t0, 4))
