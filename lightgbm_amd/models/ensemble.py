"""LightGBM text models as plain NumPy structures.

``TreeEnsemble`` parses the text format the boosters write (and that upstream
LightGBM writes: header, ``Tree=i`` blocks, ``end of trees``) into arrays, predicts
with a vectorised NumPy traversal, and tabulates the trees.  It shares no code
with the C++ parser/predictor or the HIP predictor, so the tests use it as the
independent oracle for model-file round trips and for both native predictors
(the reference keeps only predictions and throws its models away,
r/gridsearchCV.R:127, so the file format is checked against this reader
instead of against stored fixtures).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_DEFAULT_LEFT = 2      # decision_type bit 1
_MISSING_SHIFT = 2     # decision_type bits 2-3: 0 none, 1 zero, 2 NaN


@dataclass
class TreeArrays:
    num_leaves: int
    split_feature: np.ndarray
    threshold: np.ndarray
    decision_type: np.ndarray
    left_child: np.ndarray
    right_child: np.ndarray
    leaf_value: np.ndarray
    split_gain: np.ndarray = field(default_factory=lambda: np.zeros(0))
    leaf_count: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    internal_count: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    internal_value: np.ndarray = field(default_factory=lambda: np.zeros(0))
    shrinkage: float = 1.0
    # categorical nodes (decision_type bit 0): threshold = k, bitset words cat_threshold[cat_boundaries[k]:[k+1]]
    cat_boundaries: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    cat_threshold: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))

    def _in_category_set(self, nd: np.ndarray, v: np.ndarray) -> np.ndarray:
        """For categorical nodes nd and values v: is the (integer, non-negative) value in the node's set?"""
        ok = v >= 0                                   # NaN and negative codes are never in a set
        c = np.where(ok, v, 0).astype(np.int64)
        k = self.threshold[nd].astype(np.int64)
        lo, hi = self.cat_boundaries[k], self.cat_boundaries[k + 1]
        word = lo + (c >> 5)
        ok &= word < hi
        bits = self.cat_threshold[np.where(ok, word, 0)]
        return ok & (((bits >> (c & 31)) & 1) == 1)

    def predict_leaf(self, X: np.ndarray) -> np.ndarray:
        """Leaf index of every row of X (float64 [n, F])."""
        n = X.shape[0]
        if self.num_leaves <= 1:
            return np.zeros(n, dtype=np.int64)
        node = np.zeros(n, dtype=np.int64)
        active = np.ones(n, dtype=bool)
        rows = np.arange(n)
        while active.any():
            idx = rows[active]
            nd = node[idx]
            v = X[idx, self.split_feature[nd]].astype(np.float64)
            dt = self.decision_type[nd].astype(np.int64)
            missing_type = (dt >> _MISSING_SHIFT) & 3
            isnan = np.isnan(v)
            v = np.where(isnan & (missing_type != 2), 0.0, v)
            is_missing = ((missing_type == 1) & (np.abs(v) < 1e-35)) | ((missing_type == 2) & isnan)
            with np.errstate(invalid="ignore"):
                left = np.where(is_missing, (dt & _DEFAULT_LEFT) != 0, v <= self.threshold[nd])
            is_cat = (dt & 1) != 0
            if is_cat.any():
                raw = X[idx[is_cat], self.split_feature[nd[is_cat]]].astype(np.float64)
                left = left.copy()
                with np.errstate(invalid="ignore"):
                    left[is_cat] = self._in_category_set(nd[is_cat], raw)
            nxt = np.where(left, self.left_child[nd], self.right_child[nd])
            node[idx] = nxt
            active[idx] = nxt >= 0
        return ~node

    def depth(self) -> int:
        if self.num_leaves <= 1:
            return 0
        best, stack = 0, [(0, 1)]
        while stack:
            nd, d = stack.pop()
            for c in (int(self.left_child[nd]), int(self.right_child[nd])):
                if c >= 0:
                    stack.append((c, d + 1))
                else:
                    best = max(best, d)
        return best


def _block_to_dict(lines: List[str]) -> Dict[str, str]:
    out = {}
    for ln in lines:
        if "=" in ln:
            k, v = ln.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def _arr(d: Dict[str, str], key: str, dtype, n: int) -> np.ndarray:
    if n <= 0 or key not in d or not d[key]:
        return np.zeros(0, dtype=dtype)
    a = np.array(d[key].split(), dtype=np.float64 if dtype is float else dtype)
    if len(a) != n:
        raise ValueError(f"model parse: {key} has {len(a)} entries, expected {n}")
    return a


class TreeEnsemble:
    """A parsed text model: header fields + one :class:`TreeArrays` per tree."""

    def __init__(self, trees: List[TreeArrays], header: Dict[str, str], parameters: Dict[str, str],
                 average_output: bool):
        self.trees = trees
        self.header = header
        self.parameters = parameters
        self.average_output = average_output
        self.objective = header.get("objective", "regression")
        self.feature_names = header.get("feature_names", "").split()
        self.max_feature_idx = int(header.get("max_feature_idx", len(self.feature_names) - 1))
        # trees per iteration: tree t belongs to iteration t // K and class t % K
        self.num_class = int(header.get("num_tree_per_iteration", header.get("num_class", "1")))
        if self.num_class < 1 or len(trees) % self.num_class:
            raise ValueError("model parse: the number of trees is not a multiple of num_tree_per_iteration")

    # -- construction -------------------------------------------------------
    @classmethod
    def from_string(cls, text: str) -> "TreeEnsemble":
        if "end of trees" not in text:
            raise ValueError("model parse: 'end of trees' marker not found")
        body, tail = text.split("end of trees", 1)
        chunks = body.split("Tree=")
        head_lines = [ln.strip() for ln in chunks[0].splitlines() if ln.strip()]
        if not head_lines or head_lines[0] != "tree":
            raise ValueError("model parse: not a tree model")
        header = _block_to_dict(head_lines)
        average_output = "average_output" in head_lines
        trees = []
        for c in chunks[1:]:
            lines = c.splitlines()
            d = _block_to_dict(lines[1:])
            L = int(d["num_leaves"])
            n_cat = int(d.get("num_cat", "0"))
            ni = L - 1
            trees.append(TreeArrays(
                num_leaves=L,
                split_feature=_arr(d, "split_feature", np.int64, ni),
                threshold=_arr(d, "threshold", float, ni),
                decision_type=_arr(d, "decision_type", np.int64, ni),
                left_child=_arr(d, "left_child", np.int64, ni),
                right_child=_arr(d, "right_child", np.int64, ni),
                leaf_value=_arr(d, "leaf_value", float, L),
                split_gain=_arr(d, "split_gain", float, ni),
                leaf_count=_arr(d, "leaf_count", np.int64, L) if d.get("leaf_count") else np.zeros(0, dtype=np.int64),
                internal_count=_arr(d, "internal_count", np.int64, ni),
                internal_value=_arr(d, "internal_value", float, ni),
                shrinkage=float(d.get("shrinkage", "1")),
                cat_boundaries=_arr(d, "cat_boundaries", np.int64, n_cat + 1) if n_cat else np.zeros(0, dtype=np.int64),
                cat_threshold=np.asarray(d.get("cat_threshold", "").split(), dtype=np.int64) if n_cat else np.zeros(0, dtype=np.int64),
            ))
        params: Dict[str, str] = {}
        if "parameters:" in tail:
            for ln in tail.split("parameters:", 1)[1].splitlines():
                ln = ln.strip()
                if ln.startswith("[") and ln.endswith("]") and ":" in ln:
                    k, v = ln[1:-1].split(":", 1)
                    params[k.strip()] = v.strip()
        return cls(trees, header, params, average_output)

    @classmethod
    def from_file(cls, path: str) -> "TreeEnsemble":
        with open(path, "r") as f:
            return cls.from_string(f.read())

    @classmethod
    def from_booster(cls, booster) -> "TreeEnsemble":
        return cls.from_string(booster.model_to_string(num_iteration=-1))

    # -- inspection ---------------------------------------------------------
    @property
    def num_trees(self) -> int:
        return len(self.trees)

    def depths(self) -> List[int]:
        return [t.depth() for t in self.trees]

    def to_dataframe(self):
        """One row per node, like upstream's ``Booster.trees_to_dataframe``."""
        import pandas as pd
        rows = []
        for ti, t in enumerate(self.trees):
            name = lambda c: f"{ti}-S{c}" if c >= 0 else f"{ti}-L{~c}"
            parent: Dict[int, Optional[str]] = {0: None} if t.num_leaves > 1 else {~0: None}
            depth: Dict[int, int] = {0: 1} if t.num_leaves > 1 else {~0: 1}
            for nd in range(t.num_leaves - 1):
                for c in (int(t.left_child[nd]), int(t.right_child[nd])):
                    parent[c] = name(nd)
                    depth[c] = depth.get(nd, 1) + 1
            for nd in range(t.num_leaves - 1):
                f = int(t.split_feature[nd])
                dt = int(t.decision_type[nd])
                rows.append(dict(tree_index=ti, node_depth=depth.get(nd, 1), node_index=name(nd),
                                 left_child=name(int(t.left_child[nd])), right_child=name(int(t.right_child[nd])),
                                 parent_index=parent.get(nd),
                                 split_feature=self.feature_names[f] if f < len(self.feature_names) else f"Column_{f}",
                                 split_gain=float(t.split_gain[nd]) if len(t.split_gain) else np.nan,
                                 threshold=float(t.threshold[nd]), decision_type="<=",
                                 missing_direction="left" if dt & _DEFAULT_LEFT else "right",
                                 missing_type=("None", "Zero", "NaN")[(dt >> _MISSING_SHIFT) & 3],
                                 value=float(t.internal_value[nd]) if len(t.internal_value) else np.nan,
                                 count=int(t.internal_count[nd]) if len(t.internal_count) else -1))
            for lf in range(t.num_leaves):
                rows.append(dict(tree_index=ti, node_depth=depth.get(~lf, 1), node_index=name(~lf), left_child=None,
                                 right_child=None, parent_index=parent.get(~lf), split_feature=None, split_gain=np.nan,
                                 threshold=np.nan, decision_type=None, missing_direction=None, missing_type=None,
                                 value=float(t.leaf_value[lf]),
                                 count=int(t.leaf_count[lf]) if len(t.leaf_count) else -1))
        return pd.DataFrame(rows)

    # -- prediction ---------------------------------------------------------
    def predict(self, X, raw_score: bool = False, start_iteration: int = 0, num_iteration: Optional[int] = None,
                pred_leaf: bool = False) -> np.ndarray:
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] <= self.max_feature_idx:
            raise ValueError(f"data has {X.shape[1] if X.ndim == 2 else '?'} columns, the model needs {self.max_feature_idx + 1}")
        K = self.num_class
        n_iter = self.num_trees // K
        end = n_iter if num_iteration is None or num_iteration <= 0 else min(n_iter, start_iteration + num_iteration)
        use = self.trees[start_iteration * K:end * K]
        if pred_leaf:
            return np.stack([t.predict_leaf(X) for t in use], axis=1) if use else np.zeros((X.shape[0], 0), dtype=np.int64)
        out = np.zeros((X.shape[0], K), dtype=np.float64)
        for i, t in enumerate(use):
            out[:, i % K] += t.leaf_value[t.predict_leaf(X)]
        if self.average_output and use:
            out /= len(use) // K
        obj = self.objective.split()[0]
        sigmoid = 1.0
        for tok in self.objective.split()[1:]:
            if tok.startswith("sigmoid:"):
                sigmoid = float(tok.split(":", 1)[1])
        if K > 1:
            if raw_score:
                return out
            if obj == "multiclassova":
                return 1.0 / (1.0 + np.exp(-sigmoid * out))
            e = np.exp(out - out.max(axis=1, keepdims=True))
            return e / e.sum(axis=1, keepdims=True)
        out = out[:, 0]
        if raw_score:
            return out
        if obj == "binary":
            return 1.0 / (1.0 + np.exp(-sigmoid * out))
        if obj in ("cross_entropy", "xentropy"):
            return 1.0 / (1.0 + np.exp(-out))
        if obj in ("poisson", "gamma", "tweedie"):
            return np.exp(out)
        return out
