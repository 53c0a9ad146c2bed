"""Randomised CPU-oracle vs HIP-engine parity sweep.

Every seed draws a dataset shape (crossing the engine's size gates: row lines at 2^18 rows, two rounds
in flight (when asked for) at 2^19, tree-walk score update at 2^20; bags up to 2^16 rows take the whole-tree kernel and are
also trained with it switched off), a feature count (one or two 32-feature tiles),
missing values, weights and a parameter set, trains a few iterations on both engines and compares the
model text.  Regression objectives only: their gradients involve no transcendental function, so the
integer histograms - and with them the trees - are bit-identical.  The percentile-output objectives
(regression_l1, quantile, mape) also cover the device ranking of leaf residuals.

    python tools/parity_sweep.py [first_seed] [n_seeds] [max_rows] [full]

``full``: both engines with full-precision (36-bit fixed-point) histograms, ``use_quantized_grad=false``.
"""
from __future__ import annotations

import sys
import time

import numpy as np


def draw_case(seed: int, max_rows: int):
    rng = np.random.RandomState(1000 + seed)
    sizes = [n for n in (900, 6000, 45000, 270000, 530000, 1_100_000) if n <= max_rows]
    n = int(sizes[rng.randint(len(sizes))]) + int(rng.randint(0, 7))
    F = int([3, 6, 12, 28, 40][rng.randint(5)])
    if n > 300000:
        F = min(F, 28)
    p = {
        "num_leaves": int([2, 5, 15, 31, 63, 255][rng.randint(6)]),
        "min_data_in_leaf": int([1, 5, 20, 100][rng.randint(4)]),
        "max_bin": int([7, 63, 255][rng.randint(3)]),
        "boost_from_average": False,
        "objective": ["regression", "huber", "fair", "regression_l1", "quantile", "mape"][rng.randint(6)],
        "learning_rate": float([0.3, 0.1][rng.randint(2)]),
    }
    if rng.rand() < 0.3:
        p["max_depth"] = int(rng.randint(2, 8))
    if rng.rand() < 0.3:
        p["lambda_l1"] = 0.05
    if rng.rand() < 0.3:
        p["lambda_l2"] = 1.0
    if rng.rand() < 0.2:
        p["min_gain_to_split"] = 0.01
    if rng.rand() < 0.4:
        p["bagging_fraction"] = float([0.3, 0.6, 0.9][rng.randint(3)])
        p["bagging_freq"] = int(rng.randint(1, 3))
    if rng.rand() < 0.3:
        p["feature_fraction"] = 0.7
    if rng.rand() < 0.2:
        p["feature_fraction_bynode"] = 0.6
    if rng.rand() < 0.2 and "bagging_fraction" not in p:
        p["data_sample_strategy"] = "goss"
        p["top_rate"] = float([0.1, 0.3][rng.randint(2)])
        p["other_rate"] = float([0.1, 0.25][rng.randint(2)])
        p["learning_rate"] = 0.5          # two warm-up iterations, then sampled ones
    elif rng.rand() < 0.15:
        p["boosting"] = "dart"
        p["drop_rate"] = 0.5
        p["skip_drop"] = 0.0
    elif rng.rand() < 0.15:
        p["boosting"] = "rf"
        p.setdefault("bagging_fraction", 0.6)
        p.setdefault("bagging_freq", 1)
    if rng.rand() < 0.15:
        p["extra_trees"] = True
    if rng.rand() < 0.25:
        p["monotone_constraints"] = [int(v) for v in rng.randint(-1, 2, size=F)]
    if rng.rand() < 0.2:
        p["path_smooth"] = float([5.0, 50.0][rng.randint(2)])
    if rng.rand() < 0.2 and F >= 6:
        cut = int(rng.randint(2, F - 2))
        p["interaction_constraints"] = [list(range(0, cut)), list(range(cut - 1, F - 1))]   # last column in no group
    if rng.rand() < 0.15:
        p["zero_as_missing"] = True
    if rng.rand() < 0.15:
        p["feature_contri"] = [float(v) for v in np.round(0.2 + rng.rand(F) * 1.5, 2)]
    if rng.rand() < 0.15:
        p["cegb_penalty_split"] = 1e-5
        p["cegb_penalty_feature_coupled"] = [float(v) for v in np.round(rng.rand(F) * 0.5, 2)]
    opts = {"nan": rng.rand() < 0.4, "weight": rng.rand() < 0.3, "lowcard": rng.rand() < 0.4,
            "rounds": int(rng.randint(3, 7)) + (4 if p.get("boosting") == "dart" else 0), "subset": rng.rand() < 0.25,
            "categorical": rng.rand() < 0.3}
    if opts["categorical"] and rng.rand() < 0.3:
        p["max_cat_to_onehot"] = 64            # one category against the rest, on the device
    if opts["categorical"] and "monotone_constraints" in p:
        p["monotone_constraints"][F - 2 if F > 1 else 0] = 0
    if rng.rand() < 0.12:                      # (drawn last: the cases of earlier sweeps keep their settings)
        p["stochastic_rounding"] = True
        p["seed"] = int(rng.randint(1, 1000))
    return n, F, p, opts


def make_data(seed, n, F, opts):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, F)
    if opts["lowcard"]:
        X[:, F - 1] = rng.randint(0, 3, size=n)
        X[:, 0] = np.round(X[:, 0], 1)
    y = X[:, 0] * 1.5 + np.abs(X[:, 1 % F]) - 0.7 * X[:, 2 % F] * (X[:, 0] > 0) + 0.1 * rng.randn(n)
    if opts.get("categorical"):
        c = F - 2 if F > 1 else 0
        ncat = 9 if seed % 2 else 40            # 9: still above max_cat_to_onehot; 40: long sorted searches
        codes = rng.randint(0, ncat, size=n)
        X[:, c] = codes
        y = y + np.random.RandomState(7).randn(40)[codes] * 1.5
    if opts["nan"]:
        X[rng.rand(n) < 0.07, 1 % F] = np.nan
        X[rng.rand(n) < 0.01, F - 1] = np.nan
    w = (0.5 + rng.rand(n)) if opts["weight"] else None
    return X, y, w


def run_case(seed: int, max_rows: int, full: bool = False):
    import lightgbm_amd as lgb
    n, F, p, opts = draw_case(seed, max_rows)
    if full:
        p["use_quantized_grad"] = False      # the one-workgroup builds keep 16-bit records: growth rounds either way
    X, y, w = make_data(seed, n, F, opts)
    models = []
    rows = (n + 1) // 2 if opts["subset"] else n
    # small bags are grown by the whole-tree kernel: such a case runs the device engine both ways
    for dev in ("cpu", "gpu") + (("gpu-rounds",) if rows <= 65536 else ()):
        base = {"device_type": dev.split("-")[0], "verbose": -1}
        if dev == "gpu-rounds":
            base["whole_tree_max_rows"] = 0
        elif dev == "gpu" and rows <= 65536:
            base["whole_tree_max_rows"] = 65536      # explicit: also a booster that trains alone takes that path
        cats = [F - 2 if F > 1 else 0] if opts.get("categorical") else "auto"
        dp = {**base, "max_bin": p["max_bin"], "zero_as_missing": p.get("zero_as_missing", False)}
        d = lgb.Dataset(X, label=y, weight=w, params=dp, categorical_feature=cats)
        if opts["subset"]:
            d = d.construct().subset(np.arange(0, n, 2))
        b = lgb.train({**base, **p}, d, opts["rounds"])
        s = b.model_to_string()
        models.append(s[s.index("Tree=0"):s.index("end of trees")])
    return all(m == models[0] for m in models[1:]), (n, F, p, opts)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1_200_000
    full = len(sys.argv) > 4 and sys.argv[4] == "full"
    bad = 0
    for seed in range(first, first + count):
        t0 = time.time()
        ok, case = run_case(seed, max_rows, full)
        print(f"seed {seed}: {'ok' if ok else 'MISMATCH'} {time.time() - t0:.1f}s {case}", flush=True)
        bad += 0 if ok else 1
    print(f"{count - bad}/{count} identical", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    sys.exit(main())
