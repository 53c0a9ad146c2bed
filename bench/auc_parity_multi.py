"""Held-out AUC of the three learners of the headline benchmark over several independent data draws.

For every data seed the Higgs-shaped job (10.5 M x 28 training rows, 500 k held-out rows, 255 leaves / 255 bins,
lr 0.1, min_sum_hessian_in_leaf 100, 500 trees) is trained by

* ``gpu q16``   - the device engine with 16-bit fixed-point gradient histograms (one packed LDS atomic per cell),
* ``gpu full``  - the device engine with full-precision (36-bit fixed-point) histograms, ``use_quantized_grad=false``,
* ``cpu full``  - the CPU engine with the same full-precision histograms (the reference of the +-0.001 AUC bar),

and the held-out AUC / logloss of the final model plus the SHA-256 of its tree text are appended as one JSON line
per (seed, learner) to ``--out``.  ``gpu full`` and ``cpu full`` are expected to give the SAME digest on data whose
gradients involve no device-specific transcendental rounding; for the binary objective exp() differs in the last
bits between libm and the device, so their digests may differ while their metrics agree.

    python bench/auc_parity_multi.py --seeds 1 --learners "gpu q16,gpu full,cpu full" --out gpurun_out/r3/auc.jsonl
    python bench/auc_parity_multi.py --summarise gpurun_out/r3/auc.jsonl
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np                                                    # noqa: E402


LEARNERS = {
    "gpu q16": {"device_type": "gpu", "use_quantized_grad": True},
    "gpu q16 stochastic": {"device_type": "gpu", "use_quantized_grad": True, "stochastic_rounding": True},
    "gpu full": {"device_type": "gpu", "use_quantized_grad": False},
    "cpu q16": {"device_type": "cpu", "use_quantized_grad": True},
    "cpu full": {"device_type": "cpu", "use_quantized_grad": False},
}


def auc_of(y, p):
    order = np.argsort(p, kind="mergesort")
    y = y[order]
    p = p[order]
    # midranks for ties
    n = len(p)
    ranks = np.empty(n, dtype=np.float64)
    i = 0
    idx = np.flatnonzero(np.r_[True, p[1:] != p[:-1], True])
    for a, b in zip(idx[:-1], idx[1:]):
        ranks[a:b] = 0.5 * (a + b - 1) + 1.0
    npos = float(y.sum())
    nneg = n - npos
    return float((ranks[y > 0].sum() - npos * (npos + 1) / 2) / (npos * nneg))


def summarise(path):
    rows = [json.loads(l) for l in open(path) if l.strip().startswith("{")]
    by = {}
    for r in rows:
        by.setdefault(r["learner"], {})[r["seed"]] = r
    names = sorted(by)
    print("held-out AUC after %d trees, %d training rows; one column per data seed" % (rows[0]["trees"], rows[0]["rows"]))
    seeds = sorted({r["seed"] for r in rows})
    print("%-20s " % "learner" + " ".join("%9d" % s for s in seeds) + "      mean        sd   seconds")
    for n in names:
        v = [by[n][s]["auc"] for s in seeds if s in by[n]]
        t = [by[n][s]["train_seconds"] for s in seeds if s in by[n]]
        cells = " ".join(("%9.6f" % by[n][s]["auc"]) if s in by[n] else "        -" for s in seeds)
        print("%-20s %s  %9.6f %9.2e %9.1f" % (n, cells, np.mean(v), np.std(v, ddof=1) if len(v) > 1 else 0.0, np.mean(t)))
    ref = "cpu full" if "cpu full" in by else ("gpu full" if "gpu full" in by else None)
    if ref:
        for n in names:
            if n == ref:
                continue
            common = [s for s in seeds if s in by[n] and s in by[ref]]
            if not common:
                continue
            d = np.array([by[n][s]["auc"] - by[ref][s]["auc"] for s in common])
            same = sum(by[n][s]["trees_sha256"] == by[ref][s]["trees_sha256"] for s in common)
            print("%-20s - %s: mean difference %+.2e (|.| %s the 0.001 bar), largest single %.2e, sd %.2e, identical tree text on %d of %d seeds"
                  % (n, ref, d.mean(), "inside" if abs(d.mean()) <= 1e-3 else "OUTSIDE", np.abs(d).max(),
                     d.std(ddof=1) if len(d) > 1 else 0.0, same, len(common)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_500_000)
    ap.add_argument("--test-rows", type=int, default=500_000)
    ap.add_argument("--trees", type=int, default=500)
    ap.add_argument("--seeds", default="1")
    ap.add_argument("--learners", default="gpu q16,gpu full,cpu full")
    ap.add_argument("--out", default="auc_parity.jsonl")
    ap.add_argument("--objective", default="binary")
    ap.add_argument("--summarise", default="")
    args = ap.parse_args()
    if args.summarise:
        summarise(args.summarise)
        return
    import lightgbm_amd as lgb
    from lightgbm_amd.utils.datasets import make_higgs_like
    for seed in [int(s) for s in args.seeds.split(",")]:
        X, y = make_higgs_like(args.rows + args.test_rows, seed=1000 + seed, device="numpy")
        Xtr, ytr, Xte, yte = X[: args.rows], y[: args.rows], X[args.rows:], y[args.rows:]
        for name in [s.strip() for s in args.learners.split(",")]:
            if LEARNERS[name]["device_type"] == "gpu" and lgb.default_device() != "gpu":
                continue
            p = {"objective": args.objective, "num_leaves": 255, "max_bin": 255, "learning_rate": 0.1, "min_data_in_leaf": 1,
                 "min_sum_hessian_in_leaf": 100.0, "verbose": -1, "seed": seed, **LEARNERS[name]}
            t0 = time.time()
            d = lgb.Dataset(Xtr, label=ytr, params=p).construct()
            t_bin = time.time() - t0
            b = lgb.Booster(p, d)
            t0 = time.time()
            for it in range(args.trees):
                b.update()
                if (it + 1) % 25 == 0:
                    print(f"[seed {seed}] {name}: {it + 1} trees, {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
            t_train = time.time() - t0
            s = b.model_to_string()
            trees = s[s.index("Tree=0"):s.index("end of trees")]
            pred = b.predict(Xte, raw_score=True, device="cpu")
            if args.objective == "binary":
                prob = 1.0 / (1.0 + np.exp(-pred))
                ll = float(-np.mean(yte * np.log(np.clip(prob, 1e-15, 1)) + (1 - yte) * np.log(np.clip(1 - prob, 1e-15, 1))))
                rec_metric = {"auc": auc_of(yte, pred), "logloss": ll}
            else:
                rec_metric = {"auc": float("nan"), "l2": float(np.mean((pred - yte) ** 2))}
            rec = {"seed": seed, "learner": name, "rows": args.rows, "trees": args.trees, "test_rows": args.test_rows,
                   **rec_metric, "train_seconds": round(t_train, 3), "binning_seconds": round(t_bin, 3),
                   "trees_sha256": hashlib.sha256(trees.encode()).hexdigest()}
            with open(args.out, "a") as f:
                f.write(json.dumps(rec) + "\n")
            print(json.dumps(rec), flush=True)
            del b, d


if __name__ == "__main__":
    main()
