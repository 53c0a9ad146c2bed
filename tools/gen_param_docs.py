"""Writes docs/PARAMETERS.md from the native parameter table (names, aliases, defaults) plus the notes below.

    python tools/gen_param_docs.py
"""
from __future__ import annotations

import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

NOTES = {
    "boosting": "`gbdt`, `rf` (random forest: bagged trees on the initial gradients, averaged output), `dart`; `goss` is accepted and means `gbdt` with `data_sample_strategy=goss`",
    "objective": "`regression`, `regression_l1`, `huber`, `fair`, `poisson`, `quantile`, `mape`, `gamma`, `tweedie`, `binary`, `cross_entropy`, `multiclass`, `multiclassova`, `custom` (gradients from `fobj`)",
    "metric": "comma list; empty = the objective's own loss. `l2`, `rmse`, `l1`, `huber`, `fair`, `poisson`, `quantile`, `mape`, `gamma`, `gamma_deviance`, `tweedie`, `binary_logloss`, `binary_error`, `auc`, `cross_entropy`, `multi_logloss`, `multi_error`, `none`",
    "tree_learner": "`serial`, `data` (rows sharded over ranks, one integer histogram all-reduce per growth round), `feature` (every rank scans its feature tiles, winners exchanged), `voting` (rows sharded; per leaf the ranks vote for `top_k` features and only the 2 x `top_k` most voted features' histogram rows are summed)",
    "device_type": "`gpu` (HIP engine, default when a GPU is visible) or `cpu` (OpenMP engine, same numerics)",
    "num_iterations": "boosting rounds (`nrounds` in the R-style entry points)",
    "learning_rate": "shrinkage of every tree",
    "num_leaves": "leaf budget of the best-first growth",
    "max_depth": "<= 0: unlimited",
    "min_data_in_leaf": "minimum rows per child",
    "min_sum_hessian_in_leaf": "minimum hessian sum per child",
    "min_gain_to_split": "a split must improve the objective by more than this",
    "lambda_l1": "L1 regularisation of leaf outputs",
    "lambda_l2": "L2 regularisation of leaf outputs",
    "max_delta_step": "cap on |leaf output| (<= 0: none)",
    "max_bin": "bins per feature, at most 255 value bins + a missing-value bin",
    "min_data_in_bin": "minimum sampled rows per bin",
    "bin_construct_sample_cnt": "rows sampled to find the bin boundaries",
    "use_missing": "NaN gets its own bin and a learned default direction",
    "zero_as_missing": "refused (NaN is the missing value)",
    "feature_fraction": "share of the features offered to each tree",
    "feature_fraction_bynode": "share of the features offered at each node (random-forest `mtry`)",
    "feature_fraction_seed": "",
    "extra_trees": "extremely randomised trees: every node considers one random threshold per feature",
    "extra_seed": "",
    "bagging_fraction": "share of the rows in a bag (without replacement)",
    "bagging_freq": "draw a new bag every k iterations (0: no bagging)",
    "bagging_seed": "also seeds goss",
    "data_sample_strategy": "`bagging` or `goss` (keep the `top_rate` rows with the largest |g h|, sample `other_rate` of the rest and up-weight them)",
    "top_rate": "goss", "other_rate": "goss",
    "categorical_feature": "comma list of column indices holding non-negative integer category codes (the Python `Dataset(categorical_feature=...)` argument also takes names and finds pandas `category` columns); every category frequent enough gets a bin, a split sends one category left",
    "monotone_constraints": "comma list, one of -1 / 0 / +1 per column: the prediction may not increase / is free / may not decrease with that feature",
    "drop_rate": "dart: share of the existing iterations muted per iteration",
    "max_drop": "dart: at most this many (<= 0: no limit)",
    "skip_drop": "dart: chance of an iteration without muting",
    "uniform_drop": "dart: mute uniformly instead of proportionally to the trees' weights",
    "xgboost_dart_mode": "dart: xgboost's normalisation",
    "drop_seed": "dart",
    "sigmoid": "binary / multiclassova: steepness of the link",
    "num_class": "multiclass objectives: number of classes (= trees per iteration)",
    "is_unbalance": "binary: weigh the classes to equal total weight",
    "scale_pos_weight": "binary: weight multiplier of the positive class",
    "boost_from_average": "start from the label mean / log-odds / class priors",
    "alpha": "huber threshold; quantile level",
    "fair_c": "fair loss constant",
    "poisson_max_delta_step": "poisson: safeguard term in the hessian",
    "tweedie_variance_power": "tweedie: in [1, 2)",
    "num_threads": "CPU engine threads (0: OpenMP default)",
    "seed": "sets the other seeds",
    "verbose": "< 0 fatal only, 0 warnings, 1 info",
    "early_stopping_round": "stop when no validation metric improved for this many rounds",
    "first_metric_only": "early stopping looks at the first metric only",
    "metric_freq": "evaluate every k iterations",
    "is_provide_training_metric": "record the training metric too",
    "snapshot_freq": "write `<output_model>.snapshot_iter_<k>` every k iterations (resume with `init_model`)",
    "output_model": "CLI: model file written by `task=train`",
    "input_model": "CLI: model file read by `task=predict` / continued training",
    "num_machines": "informational; the communicator defines the world",
    "gpu_device_id": "informational; the current torch device is used",
    "whole_tree_max_rows": "device engine: a tree over at most this many in-bag rows (<= 65536) is grown by ONE workgroup in one launch - the whole best-first loop on the device, no host round trips - so many small boosters (CV folds, grid rows) run side by side, whole chunks of iterations without the host; taken when several boosters are advanced together (cv, grid_search_cv, Booster.update_group) or when this parameter is given explicitly - a booster training alone is faster with batched rounds over the whole device; 0 grows every tree in batched rounds",
    "pos_bagging_fraction": "binary objective: share of the positive rows kept by balanced bagging (with neg_bagging_fraction; needs bagging_freq > 0)",
    "neg_bagging_fraction": "binary objective: share of the negative rows kept by balanced bagging",
    "max_cat_to_onehot": "a categorical feature with at most this many categories is split one category against the rest (on the device); above it, sets of categories sorted by gradient statistics are searched",
    "cat_smooth": "category-set search: smoothing added to a category's hessian sum before sorting by g / h",
    "cat_l2": "extra L2 regularisation of a split on a category set",
    "max_cat_threshold": "category-set search: most categories a set may hold",
    "min_data_per_group": "category-set search: fewest rows of a category group",
    "feature_contri": "per feature (raw column order): factor on the gain of its splits",
    "cegb_tradeoff": "cost-effective gradient boosting: factor on all penalties",
    "cegb_penalty_split": "cost-effective gradient boosting: penalty per split, times the rows of the leaf",
    "cegb_penalty_feature_coupled": "cost-effective gradient boosting: per feature, a penalty paid by the first split on that feature in the whole model",
    "cegb_penalty_feature_lazy": "the per-row lazy penalty is not implemented: a list with a non-zero entry is refused",
    "use_quantized_grad": "true: 16-bit fixed-point gradient histograms (one packed 64-bit LDS atomic per row and feature on the device); false: full precision - the float gradients as 36-bit fixed point, summed exactly in 64-bit cells (two LDS atomics per row and feature; growth rounds only, the one-workgroup builds of small boosters keep 16 bits). Either way both engines, and any number of ranks, grow bit-identical trees",
    "quant_train_renew_leaf": "leaf values are recomputed from exact 36-bit sums of the float gradients of each leaf's rows (both engines; off: the values the 16-bit search produced)",
    "path_smooth": "> 0: a child's output is pulled towards its parent's, the more so the fewer rows it holds",
    "interaction_constraints": "groups of raw column indices, e.g. [0,1,2],[2,3]: a branch may only combine features of one group",
    "max_bin_by_feature": "per raw column: its own max_bin",
    "lambdarank_truncation_level": "lambdarank: only pairs with a document in the top-k of the current ranking contribute",
    "lambdarank_norm": "lambdarank: normalise a query's gradients by the sum of its pair weights",
    "label_gain": "ranking: gain of each integer relevance grade (default 2^grade - 1)",
    "objective_seed": "rank_xendcg: seed of the per-iteration noise",
    "eval_at": "ndcg / map cut-offs (alias ndcg_eval_at)",
    "label_column": "command line, text data files: index of the label column",
    "header": "command line, text data files: the first line holds column names",
    "max_batch_leaves": "leaves split per device round (0: as many as the speculative growth wants; 1 reproduces one-leaf-at-a-time growth)",
    "profile": "per-phase HIP-event timers, read with `Booster.profile()`",
    "locality_order": "device engine, data of one feature tile with at least 2^20 rows and no row subset: the booster keeps private copies of the binned matrices in the order of a coarse key over the features whose bins explain the label best, so that the rows of a deep leaf sit in runs instead of being spread over the table (same trees, bit for bit; the random streams of bagging / GOSS / stochastic rounding stay keyed by the caller's row numbers)",
    "device_bin_find": "find bin boundaries on the device for GPU datasets (default on)",
    "deterministic": "always on: integer histograms make every run, engine and rank count give the same trees",
    "verbosity": "alias kept for LightGBM configs; see `verbose`",
    "data_random_seed": "seed of the row sample for bin finding",
    "feature_pre_filter": "accepted for compatibility; single-valued features are always dropped",
    "iteration_offset": "first iteration index (tree-parallel forests: keeps every iteration's random streams)",
    "top_k": "voting-parallel learner: features each rank votes for per leaf (both engines)",
    "forcedsplits_filename": "JSON file of splits every tree starts with (`feature`, `threshold`, `left`, `right`). A forced split is made whatever `min_data_in_leaf`, `min_sum_hessian_in_leaf` and its gain say (a child may even be empty: its value is then 0), rows with a missing value go right, and forcing ends on a branch at a categorical or unused feature, a threshold beyond the feature's last value bin, or the depth limit",
    "early_stopping_min_delta": "an iteration counts as an improvement only if it beats the best score by this much",
    "multi_error_top_k": "`multi_error` counts a row as right when its label is among the k best classes",
    "auc_mu_weights": "`auc_mu`: num_class x num_class misclassification weights, row-major",
    "saved_feature_importance_type": "importance block of saved models: 0 split counts, 1 total gains",
    "monotone_constraints_method": "`basic`, `intermediate`, `advanced` accepted; one method: output ranges with midpoint separation",
    "force_col_wise": "accepted for compatibility (no effect: one histogram layout per data shape)",
    "force_row_wise": "accepted for compatibility (no effect)",
    "histogram_pool_size": "accepted for compatibility (no effect: the pool holds every leaf's histogram)",
    "gpu_platform_id": "accepted for compatibility (no effect)",
    "gpu_use_dp": "true selects full-precision histograms (`use_quantized_grad=false`) unless `use_quantized_grad` is given explicitly",
    "num_gpu": "accepted for compatibility; GPUs are used one process each (`parallel/`)",
    "two_round": "accepted for compatibility (no effect)",
    "precise_float_parser": "accepted for compatibility (no effect)",
    "enable_bundle": "accepted for compatibility (no effect: dense storage, no feature bundling)",
    "num_grad_quant_bins": "no effect (a warning says so): the quantised form always has 16 bits, see `use_quantized_grad`",
    "stochastic_rounding": "16-bit histograms: false (default) rounds gradients to nearest; true rounds floor(s + u) with u uniform in [0, 1) from the counter-based generator keyed by (`seed`, tree, row) - unbiased, reproducible, the same on both engines (growth rounds only)",
    "forcedbins_filename": "JSON list of `{\"feature\": column, \"bin_upper_bound\": [...]}`: bin bounds those features must have; the remaining bins are spread over the intervals between them by row count",
    "reg_sqrt": "`objective=regression`: fit sign(y) sqrt|y| and square the predictions back; metrics on the original scale",
    "pred_early_stop": "prediction (binary / multiclass): stop adding trees for a row once its margin is wide enough",
    "pred_early_stop_freq": "... checked every this many iterations",
    "pred_early_stop_margin": "... against this margin (binary: 2|score|; multiclass: best minus second-best raw score)",
}


def main():
    import numpy as np
    import lightgbm_amd as lgb
    from lightgbm_amd import _lib
    lib, _ = _lib.load()
    table = json.loads(lib.lga_alias_table_json().decode())
    X = np.random.RandomState(0).randn(200, 3)
    b = lgb.train({"device_type": "cpu", "verbose": -1}, lgb.Dataset(X, label=X[:, 0]), 1)
    txt = b.model_to_string()
    block = txt[txt.index("parameters:"):txt.index("end of parameters")]
    defaults = {}
    for ln in block.splitlines():
        if ln.startswith("[") and ln.endswith("]") and ":" in ln:
            k, v = ln[1:-1].split(":", 1)
            defaults[k.strip()] = v.strip()
    defaults["verbose"] = "1"
    defaults["device_type"] = "gpu when visible, else cpu"
    lines = ["# Parameters", "",
             "Generated by `tools/gen_param_docs.py` from the native parameter table (`csrc/config.cpp`). Names and",
             "aliases follow LightGBM; unknown keys produce a warning, out-of-range values an error. Parameters are",
             "given as a dict (`params`), as keyword arguments of the R-style entry points, or as a one-row data frame",
             "(a parameter-grid row: every value may arrive as a double).", "",
             "| parameter | aliases | default | notes |", "|---|---|---|---|"]
    for name, aliases in table.items():
        d = defaults.get(name, "")
        try:
            d = repr(float(d)) if "." in d or "e" in d else d
            d = d[:-2] if d.endswith(".0") else d
        except ValueError:
            pass
        lines.append(f"| `{name}` | {', '.join('`%s`' % a for a in aliases)} | {d or '(empty)'} | {NOTES.get(name, '')} |")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "docs", "PARAMETERS.md")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    missing = [n for n in table if n not in NOTES]
    print(f"wrote {out}: {len(table)} parameters" + (f"; without a note: {missing}" if missing else ""))


if __name__ == "__main__":
    main()
