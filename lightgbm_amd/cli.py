"""Command line front end: ``python -m lightgbm_amd.cli task=train data=train.csv ...``.

Mirrors the ``key=value`` / ``config=file`` style of the LightGBM application:

* ``task=train``    data=<csv> [valid=<csv>] label_column=<idx> output_model=<path> + any training parameter
* ``task=predict``  data=<csv> input_model=<path> output_result=<path>
* ``task=convert_model``  input_model=<path> convert_model=<json path>   (dumps the model as JSON)
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


def _load_csv(path: str, label_column: int, has_header: bool):
    arr = np.loadtxt(path, delimiter=",", skiprows=1 if has_header else 0, ndmin=2)
    if label_column < 0:
        return arr, None
    y = arr[:, label_column]
    X = np.delete(arr, label_column, axis=1)
    return X, y


def main(argv=None) -> int:
    from . import Booster, Dataset, train
    kv = _parse(list(sys.argv[1:] if argv is None else argv))
    task = kv.pop("task", "train")
    label_column = int(kv.pop("label_column", "0"))
    header = kv.pop("header", "false").lower() in ("true", "1")
    if task == "train":
        X, y = _load_csv(kv.pop("data"), label_column, header)
        valid = kv.pop("valid", None)
        out = kv.pop("output_model", "LightGBM_model.txt")
        rounds = int(kv.pop("num_iterations", kv.pop("num_trees", kv.pop("num_boost_round", "100"))))
        params = {k: v for k, v in kv.items()}
        d = Dataset(X, label=y, params=params)
        valid_sets = None
        if valid:
            vx, vy = _load_csv(valid, label_column, header)
            valid_sets = [d.create_valid(vx, label=vy)]
        b = train(params, d, num_boost_round=rounds, valid_sets=valid_sets, verbose_eval=bool(valid_sets))
        b.save_model(out)
        print(f"finished training: {b.num_trees()} trees -> {out}")
    elif task == "predict":
        b = Booster(model_file=kv.pop("input_model"))
        X, _ = _load_csv(kv.pop("data"), int(kv.pop("label_column_predict", str(label_column))), header)
        p = b.predict(X, raw_score=kv.pop("predict_raw_score", "false").lower() in ("true", "1"))
        np.savetxt(kv.pop("output_result", "LightGBM_predict_result.txt"), p)
    elif task == "convert_model":
        b = Booster(model_file=kv.pop("input_model"))
        with open(kv.pop("convert_model", "model.json"), "w") as f:
            json.dump(b.dump_model(), f)
    else:
        raise SystemExit(f"unknown task {task!r}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
