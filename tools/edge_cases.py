"""Degenerate and extreme inputs: constant labels and features, all-missing columns, one to three rows, more
leaves than rows, two bins, zero weights, 1e30-scaled values, infinities, single-class targets, 4095 leaves, ...

    python tools/edge_cases.py cpu          # every case trains and predicts finite values
    python tools/edge_cases.py gpu          # ... on the HIP engine, and grows the CPU engine's trees

Run the GPU form under ``LGA_DEBUG_LIB=1`` first: the index-checking build turns a bad access into an error."""
from __future__ import annotations

import sys

import numpy as np


def cases():
    rng = np.random.RandomState(0)
    X = rng.randn(500, 5)
    y = X[:, 0] + 0.1 * rng.randn(500)
    tiny = {"min_data_in_leaf": 1, "min_data_in_bin": 1}
    yield "constant label", X, np.ones(500), {}, 3, {}
    yield "constant feature", np.c_[X, np.full(500, 3.0)], y, {}, 3, {}
    Xn = X.copy(); Xn[:, 2] = np.nan
    yield "all-NaN feature", Xn, y, {}, 3, {}
    Xn2 = X.copy(); Xn2[::2, 1] = np.nan
    yield "half NaN", Xn2, y, {"min_data_in_leaf": 1}, 3, {}
    yield "three rows", X[:3], y[:3], tiny, 3, {}
    yield "one row", X[:1], y[:1], tiny, 3, {}
    yield "fewer rows than leaves", X[:20], y[:20], {"num_leaves": 255, **tiny}, 3, {}
    yield "two bins", X, y, {"max_bin": 2}, 3, {}
    yield "three bins, two leaves", X, y, {"max_bin": 3, "num_leaves": 2}, 3, {}
    yield "all weights zero", X, y, {}, 3, {"weight": np.zeros(500)}
    yield "half of the weights zero", X, y, {}, 3, {"weight": (rng.rand(500) > 0.5).astype(float)}
    yield "values times 1e30", X * 1e30, y, {}, 3, {}
    yield "labels times 1e20", X, y * 1e20, {}, 3, {}
    yield "labels times 1e-30", X, y * 1e-30, {}, 3, {}
    yield "infinities in X", np.where(rng.rand(500, 5) < 0.01, np.inf, X), y, {}, 3, {}
    yield "binary, one class only", X, np.ones(500), {"objective": "binary"}, 3, {}
    big = rng.randn(20000, 3)
    yb = big[:, 0] * big[:, 1]
    yield "4095 leaves", big, yb, {"num_leaves": 4095, "min_data_in_leaf": 1}, 2, {}
    yield "two leaves", big, yb, {"num_leaves": 2}, 3, {}
    yield "max_depth 1", big, yb, {"max_depth": 1}, 3, {}
    yield "min_data_in_leaf above n / 2", X, y, {"min_data_in_leaf": 400}, 3, {}
    yield "min_gain_to_split never met", X, y, {"min_gain_to_split": 1e12}, 3, {}
    yield "learning rate 1e-300", X, y, {"learning_rate": 1e-300}, 3, {}
    yield "one feature", X[:, :1], y, {}, 3, {}
    yield "65 features (three tiles)", rng.randn(300, 65), rng.randn(300), {"min_data_in_leaf": 2}, 3, {}
    yield "five distinct rows", np.repeat(X[:5], 100, axis=0), np.repeat(y[:5], 100), {"min_data_in_leaf": 1}, 3, {}
    yield "bagging 1 %", X, y, {"bagging_fraction": 0.01, "bagging_freq": 1, "min_data_in_leaf": 1}, 3, {}
    yield "feature_fraction 1 %", X, y, {"feature_fraction": 0.01}, 3, {}
    yield "integer matrix", (X * 10).astype(np.int32), y, {}, 3, {}
    yield "float32, column-major", np.asfortranarray(X.astype(np.float32)), y, {}, 3, {}
    yield "multiclass with an empty class", X, (rng.rand(500) > 0.5).astype(float) * 2, \
        {"objective": "multiclass", "num_class": 3, "boost_from_average": False}, 1, {}
    yield "goss on 50 rows", X[:50], y[:50], {"data_sample_strategy": "goss", "learning_rate": 0.5, "min_data_in_leaf": 1}, 6, {}
    yield "dart muting everything", X, y, {"boosting": "dart", "drop_rate": 1.0, "skip_drop": 0.0}, 6, {}
    yield "l1", X, y, {"objective": "regression_l1"}, 3, {}
    yield "quantile 0.999", X, y, {"objective": "quantile", "alpha": 0.999}, 3, {}
    yield "monotone on every feature", X, y, {"monotone_constraints": [1, -1, 1, -1, 1]}, 3, {}

    # ranking and categorical corner cases (dk carries group / categorical_feature)
    Xr = rng.randn(400, 4)
    rel = np.clip(np.round(Xr[:, 0] + 1.5), 0, 4)
    yield "ranking: queries of one document", Xr, rel, {"objective": "lambdarank", **tiny}, 3, {"group": np.ones(400, dtype=int)}
    yield "ranking: a single query", Xr, rel, {"objective": "lambdarank", **tiny}, 3, {"group": [400]}
    yield "ranking: no relevant document anywhere", Xr, np.zeros(400), {"objective": "lambdarank", **tiny}, 3, {"group": [100] * 4}
    yield "ranking: every label equal", Xr, np.full(400, 2.0), {"objective": "rank_xendcg", **tiny}, 3, {"group": [50] * 8}
    big = rng.randn(6000, 3)
    yield ("ranking: a query beyond the device kernels' LDS budget (regression objective: host metrics only)", big,
           np.clip(np.round(big[:, 0] + 1.5), 0, 4), {"objective": "regression", "metric": "ndcg", "eval_at": [5]}, 2,
           {"group": [5000, 1000]})
    Xc = rng.randn(600, 3)
    Xc[:, 1] = 7.0
    yield "categorical: one category", Xc, Xc[:, 0], {**tiny}, 3, {"categorical_feature": [1]}
    Xc2 = rng.randn(600, 3)
    Xc2[:, 1] = np.arange(600)
    yield "categorical: every row its own category", Xc2, Xc2[:, 0], {**tiny}, 3, {"categorical_feature": [1]}
    Xc3 = rng.randn(600, 3)
    Xc3[:, 1] = rng.randint(-3, 4, size=600)
    yield "categorical: negative codes", Xc3, Xc3[:, 0] + (Xc3[:, 1] == 2), {**tiny}, 3, {"categorical_feature": [1]}
    Xc4 = rng.randn(3000, 2)
    Xc4[:, 1] = rng.randint(0, 200, size=3000)
    yield ("categorical: 200 categories, tiny groups", Xc4, Xc4[:, 0] + np.sin(Xc4[:, 1]),
           {"min_data_in_leaf": 1, "min_data_per_group": 1, "cat_smooth": 0, "cat_l2": 0, "max_cat_threshold": 128}, 3,
           {"categorical_feature": [1]})


def train_case(case, device: str):
    import lightgbm_amd as lgb
    name, X, y, p, rounds, dk = case
    base = {"device_type": device, "verbose": -1}
    dparams = {**base, **{k: v for k, v in p.items() if k in ("max_bin", "min_data_in_bin")}}
    d = lgb.Dataset(X, label=y, params=dparams, **dk)
    b = lgb.train({**base, **p}, d, rounds, valid_sets=[d] if "group" in dk else None, verbose_eval=False)
    pred = b.predict(np.asarray(X, dtype=np.float64))
    assert np.all(np.isfinite(pred)), f"{name}: non-finite prediction"
    text = b.model_to_string()
    return text[text.index("Tree=0"):text.index("end of trees")]


def main():
    device = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    bad = 0
    for case in cases():
        try:
            trees = train_case(case, device)
            if device != "cpu" and trees != train_case(case, "cpu"):
                raise AssertionError("trees differ from the CPU engine's")
            print(f"{case[0]}: ok", flush=True)
        except Exception as e:   # noqa: report every case
            bad += 1
            print(f"{case[0]}: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
    print("all cases passed" if not bad else f"{bad} cases failed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    sys.exit(main())
