"""Crosses objectives (regression - plain and reg_sqrt -, binary, multiclass, the two ranking objectives, L1, poisson) with boosting / sampling
variants and the tree-shape options (category sets, monotone and interaction constraints, path smoothing, ...)
on one small dataset with categorical columns, zeros and NaNs: every combination has to train, predict finite
values, keep its training scores equal to the predictor's and survive the model file.

    python tools/combo_sweep.py [cpu|gpu]
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import lightgbm_amd as lgb  # noqa: E402

rng = np.random.RandomState(0)
sizes = rng.randint(5, 30, size=200); n = int(sizes.sum())
codes = rng.randint(0, 30, size=n)
X = np.column_stack([rng.randn(n, 4), codes, rng.randint(0, 3, size=n)]).astype(float)
X[rng.rand(n) < 0.1, 1] = 0.0
X[rng.rand(n) < 0.05, 2] = np.nan
rel = X[:, 0] + rng.randn(30)[codes] + 0.3 * rng.randn(n)
yr = np.clip(np.round(rel + 1.5), 0, 4)
yc = (rel > 0).astype(float)
ym = np.clip(np.round(rel + 1), 0, 2)
base = {"device_type": sys.argv[1] if len(sys.argv) > 1 else "cpu", "verbose": -1, "num_leaves": 15, "min_data_in_leaf": 5}
cases = []
for obj, y, extra in (("lambdarank", yr, {}), ("rank_xendcg", yr, {}), ("regression", rel, {}), ("binary", yc, {}),
                      ("multiclass", ym, {"num_class": 3}), ("regression_l1", rel, {}), ("poisson", np.abs(rel), {}),
                      ("regression", rel * np.abs(rel), {"reg_sqrt": True})):
    for mod in ({}, {"data_sample_strategy": "goss", "learning_rate": 0.5}, {"boosting": "dart"},
                {"boosting": "rf", "bagging_fraction": 0.6, "bagging_freq": 1}, {"bagging_fraction": 0.5, "bagging_freq": 2, "feature_fraction": 0.6},
                {"path_smooth": 5.0, "monotone_constraints": [1, 0, 0, 0, 0, 0]}, {"interaction_constraints": [[0, 1, 4], [2, 3, 5]], "extra_trees": True},
                {"zero_as_missing": True, "max_cat_to_onehot": 64}, {"feature_fraction_bynode": 0.5, "max_depth": 4, "lambda_l1": 0.1, "cat_l2": 1.0, "min_data_per_group": 20}):
        cases.append((obj, y, {**extra, **mod}))
bad = 0
for obj, y, p in cases:
    try:
        grp = sizes if obj in ("lambdarank", "rank_xendcg") else None
        d = lgb.Dataset(X, label=y, group=grp, categorical_feature=[4, 5], params={**base, **p})
        b = lgb.train({**base, "objective": obj, **p}, d, 6, valid_sets=[d], verbose_eval=False, keep_training_booster=True)
        pr = b.predict(X)
        assert np.all(np.isfinite(pr)), "non-finite predictions"
        raw = b.predict(X, raw_score=True)
        inner = b._inner_predict(-1, raw=True)
        if p.get("boosting") != "dart":
            assert np.allclose(inner, raw, atol=1e-8), f"inner != predict ({np.abs(inner - raw).max()})"
        b2 = lgb.Booster(model_str=b.model_to_string())
        assert np.array_equal(b2.predict(X), pr), "round trip"
    except Exception as e:
        bad += 1
        print("FAIL", obj, p, "->", repr(e)[:300])
print(f"{len(cases) - bad}/{len(cases)} combinations fine")
sys.exit(1 if bad else 0)
