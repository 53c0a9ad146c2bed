"""Command line front end: ``python -m lightgbm_amd.cli task=train data=train.csv ...``.

Mirrors the ``key=value`` / ``config=file`` style of the LightGBM application:

* ``task=train``    data=<file> [valid=<file>[,<file>...]] label_column=<idx> output_model=<path>
                    [input_model=<path> to continue training] + any training parameter
* ``task=predict``  data=<file> input_model=<path> output_result=<path>
                    [predict_raw_score / predict_leaf_index / predict_contrib = true]
* ``task=refit``    data=<file> input_model=<path> output_model=<path> [refit_decay_rate=0.9]
* ``task=save_binary``  data=<file> [output: <file>.bin]      (binned dataset cache, loadable as data=<file>.bin)
* ``task=convert_model``  input_model=<path> convert_model=<json path>   (dumps the model as JSON)

Data files are CSV, TSV or LibSVM (``label idx:value ...``), told apart by their first line; ``header=true``
skips a header line.  Like the LightGBM application, side files next to a data file are picked up when they
exist: ``<file>.weight`` (one weight per row), ``<file>.query`` (one query size per line, for ranking) and
``<file>.init`` (one initial score per row).  ``weight_column=<idx>`` / ``group_column=<idx>`` take weights / query ids
from columns of a CSV / TSV file instead and ``ignore_column=<idx>,<idx>`` drops columns (indices count the columns
that remain once the label is taken out, as upstream does).
"""
from __future__ import annotations

import json
import sys
from typing import Dict, List

import numpy as np


def _parse(argv: List[str]) -> Dict[str, str]:
    kv: Dict[str, str] = {}
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got {a!r}")
        k, v = a.split("=", 1)
        kv[k.strip()] = v.strip()
    if "config" in kv:
        with open(kv.pop("config")) as f:
            for line in f:
                line = line.split("#")[0].strip()
                if line and "=" in line:
                    k, v = line.split("=", 1)
                    kv.setdefault(k.strip(), v.strip())
    return kv


def _load_table(path: str, label_column: int, has_header: bool, num_features: int = 0):
    """(X, y) from a CSV / TSV / LibSVM file; y is None when label_column < 0 (CSV / TSV only)."""
    with open(path) as f:
        lines = [ln for ln in (line.strip() for line in f) if ln]
    if has_header and lines:
        lines = lines[1:]
    if not lines:
        raise SystemExit(f"{path}: no data rows")
    first = lines[0]
    tokens = first.replace(",", " ").replace("\t", " ").split()
    if any(":" in t for t in tokens[1:]) or (len(tokens) == 1 and "," not in first and "\t" not in first):
        # LibSVM: label then index:value pairs (indices count from 0, like LightGBM reads them)
        rows, ys, width = [], [], num_features
        for ln in lines:
            parts = ln.split()
            ys.append(float(parts[0]))
            pairs = [(int(k), float(v)) for k, v in (t.split(":", 1) for t in parts[1:] if not t.startswith("qid"))]
            rows.append(pairs)
            if pairs:
                width = max(width, max(k for k, _ in pairs) + 1)
        X = np.zeros((len(rows), width), dtype=np.float64)
        for i, pairs in enumerate(rows):
            for k, v in pairs:
                X[i, k] = v
        return X, np.asarray(ys, dtype=np.float64)
    sep = "\t" if "\t" in first else ","
    conv = lambda t: float("nan") if t.strip().lower() in ("", "na", "nan", "null") else float(t)
    arr = np.array([[conv(t) for t in ln.split(sep)] for ln in lines], dtype=np.float64)
    if label_column < 0:
        return arr, None
    return np.delete(arr, label_column, axis=1), arr[:, label_column]


def _side_file(path: str, suffix: str):
    import os
    p = path + suffix
    return np.loadtxt(p, ndmin=1) if os.path.exists(p) else None


def _column_list(spec: str) -> List[int]:
    """"3" / "0,2,5" -> indices.  As upstream counts them: over the columns that remain once the label is taken out."""
    if not spec:
        return []
    if spec.startswith("name:"):
        raise SystemExit("columns are selected by index here (weight_column=3), not by name")
    return [int(t) for t in spec.split(",") if t.strip() != ""]


def _take_columns(X: np.ndarray, columns: Dict[str, str]):
    """Splits the weight / query-id / ignored columns off a feature matrix: (X, weight or None, query sizes or None)."""
    w_col = _column_list(columns.get("weight_column", ""))
    g_col = _column_list(columns.get("group_column", ""))
    drop = set(_column_list(columns.get("ignore_column", "")))
    if len(w_col) > 1 or len(g_col) > 1:
        raise SystemExit("weight_column and group_column name one column each")
    for c in w_col + g_col + sorted(drop):
        if not 0 <= c < X.shape[1]:
            raise SystemExit(f"column {c} does not exist (the data has {X.shape[1]} columns besides the label)")
    weight = X[:, w_col[0]].copy() if w_col else None
    group = None
    if g_col:
        ids = X[:, g_col[0]]
        # rows of a query are consecutive: a new query starts wherever the id changes
        starts = np.flatnonzero(np.concatenate([[True], ids[1:] != ids[:-1]]))
        group = np.diff(np.concatenate([starts, [len(ids)]]))
    drop.update(w_col + g_col)
    if drop:
        X = np.delete(X, sorted(drop), axis=1)
    return X, weight, group


def _dataset(path: str, label_column: int, header: bool, params, reference=None, columns=None):
    """A Dataset from a data file (+ its .weight / .query / .init side files, or the weight / query-id columns named
    by weight_column / group_column; ignore_column drops columns), or from a binned cache."""
    from . import Dataset
    if path.endswith(".bin"):
        return Dataset(path, params=params), None      # the cache holds label / weight / init_score / query sizes
    X, y = _load_table(path, label_column, header, reference.num_feature() if reference is not None else 0)
    weight, group = _side_file(path, ".weight"), _side_file(path, ".query")
    if columns:
        X, w_in, g_in = _take_columns(X, columns)
        weight = w_in if w_in is not None else weight
        group = g_in if g_in is not None else group
    if reference is not None and X.shape[1] < reference.num_feature():      # LibSVM rows without the last columns
        X = np.hstack([X, np.zeros((X.shape[0], reference.num_feature() - X.shape[1]))])
    return Dataset(X, label=y, weight=weight, group=group, init_score=_side_file(path, ".init"), params=params,
                   reference=reference), X


def _flag(kv, key: str) -> bool:
    return kv.pop(key, "false").lower() in ("true", "1")


def main(argv=None) -> int:
    from . import Booster, train
    kv = _parse(list(sys.argv[1:] if argv is None else argv))
    task = kv.pop("task", "train")
    label_column = int(kv.pop("label_column", "0"))
    header = _flag(kv, "header")
    columns = {k: kv.pop(k) for k in ("weight_column", "group_column", "ignore_column") if k in kv}
    if task == "train":
        data = kv.pop("data")
        valid = kv.pop("valid", kv.pop("valid_data", None))
        out = kv.pop("output_model", "LightGBM_model.txt")
        init = kv.pop("input_model", None)
        rounds = int(kv.pop("num_iterations", kv.pop("num_trees", kv.pop("num_boost_round", "100"))))
        params = {k: v for k, v in kv.items()}
        d, _ = _dataset(data, label_column, header, params, columns=columns)
        if init:
            d.free_raw_data = False
        valid_sets = [_dataset(v, label_column, header, params, reference=d, columns=columns)[0] for v in valid.split(",")] if valid else None
        b = train(params, d, num_boost_round=rounds, valid_sets=valid_sets, verbose_eval=bool(valid_sets), init_model=init)
        b.save_model(out)
        print(f"finished training: {b.num_trees()} trees -> {out}")
    elif task == "predict":
        b = Booster(model_file=kv.pop("input_model"))
        X, _ = _load_table(kv.pop("data"), int(kv.pop("label_column_predict", str(label_column))), header, b.num_feature())
        if columns:
            X = _take_columns(X, columns)[0]
        n_it = int(kv.pop("num_iteration_predict", "-1"))
        p = b.predict(X, raw_score=_flag(kv, "predict_raw_score"), pred_leaf=_flag(kv, "predict_leaf_index"),
                      pred_contrib=_flag(kv, "predict_contrib"), num_iteration=n_it if n_it > 0 else None,
                      start_iteration=int(kv.pop("start_iteration_predict", "0")),
                      pred_early_stop=_flag(kv, "pred_early_stop"),
                      pred_early_stop_freq=int(kv.pop("pred_early_stop_freq", "10")),
                      pred_early_stop_margin=float(kv.pop("pred_early_stop_margin", "10.0")))
        np.savetxt(kv.pop("output_result", "LightGBM_predict_result.txt"), p, fmt="%.17g", delimiter="\t")
    elif task == "refit":
        b = Booster(model_file=kv.pop("input_model"))
        path = kv.pop("data")
        X, y = _load_table(path, label_column, header, b.num_feature())
        out = kv.pop("output_model", "LightGBM_model.txt")
        b.refit(X, y, decay_rate=float(kv.pop("refit_decay_rate", "0.9")), weight=_side_file(path, ".weight")).save_model(out)
        print(f"finished refitting -> {out}")
    elif task == "save_binary":
        path = kv.pop("data")
        d, _ = _dataset(path, label_column, header, {k: v for k, v in kv.items()}, columns=columns)
        d.save_binary(path + ".bin")
        print(f"saved {path}.bin")
    elif task == "convert_model":
        b = Booster(model_file=kv.pop("input_model"))
        with open(kv.pop("convert_model", "model.json"), "w") as f:
            json.dump(b.dump_model(), f)
    else:
        raise SystemExit(f"unknown task {task!r}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
