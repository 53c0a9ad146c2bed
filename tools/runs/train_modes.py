import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lightgbm_amd as lgb
from lightgbm_amd.utils.datasets import make_diamonds_like
X, y = make_diamonds_like(45840, seed=0)
for leaves in (31, 127):
    for name, extra in (("chunks", {"whole_tree_max_rows": 65536}), ("whole trees, host between", {"__nochunk": 1, "whole_tree_max_rows": 65536}), ("default train()", {}), ("batched rounds", {"whole_tree_max_rows": 0})):
        p = {"objective": "regression", "learning_rate": 0.1, "num_leaves": leaves, "verbose": -1, **{k: v for k, v in extra.items() if not k.startswith("__")}}
        d = lgb.Dataset(X, label=y, params=p).construct()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            if "__nochunk" in extra:
                b = lgb.Booster(p, d)
                for _ in range(200):
                    b.update()
            else:
                b = lgb.train(p, d, 200, keep_training_booster=True)
            b._inner_predict(-1)
            best = min(best, time.perf_counter() - t)
        print(json.dumps({"leaves": leaves, "mode": name, "seconds_200_iterations": round(best, 4)}), flush=True)
