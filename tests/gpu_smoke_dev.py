"""Development smoke script for the first GPU run (not a pytest file)."""
import time
import numpy as np
import torch
import lightgbm_amd as lgb
from lightgbm_amd import _lib

print("hip:", _lib.is_hip(), torch.cuda.get_device_name(0))
rng = np.random.RandomState(0)
N, F = 20000, 8
X = rng.randn(N, F)
y = X[:, 0] * 2 + np.sin(X[:, 1] * 3) + 0.5 * X[:, 2] * X[:, 3] + 0.1 * rng.randn(N)
params = {"learning_rate": 0.1, "objective": "regression", "verbose": 0}
ds_c = lgb.Dataset(X[:15000], label=y[:15000], params={"device_type": "cpu"})
bc = lgb.train({**params, "device_type": "cpu"}, ds_c, num_boost_round=20)
ds_g = lgb.Dataset(X[:15000], label=y[:15000], params={"device_type": "gpu"})
t = time.time()
bg = lgb.train({**params, "device_type": "gpu"}, ds_g, num_boost_round=20)
print("gpu train", time.time() - t)
pc = bc.predict(X[15000:]); pg = bg.predict(X[15000:])
print("rmse cpu", np.sqrt(np.mean((pc - y[15000:]) ** 2)), "gpu", np.sqrt(np.mean((pg - y[15000:]) ** 2)))
print("max |cpu-gpu| pred", np.abs(pc - pg).max())
sc, sg = bc.model_to_string(), bg.model_to_string()
tc = sc[sc.index("Tree=0"):sc.index("end of trees")]; tg = sg[sg.index("Tree=0"):sg.index("end of trees")]
print("identical trees:", tc == tg)
pgd = bg.predict(X[15000:], device="gpu")
print("gpu predictor max diff", np.abs(pgd - pg).max())
