"""Dataset and Booster: the Python face of the native core.

Mirrors the calls the reference scripts make through the R binding -
``lgb.Dataset(matrix, label=)`` (r/gridsearchCV.R:52-53), ``predict(booster,
matrix)`` (r/gridsearchCV.R:63) and staged prediction with the first k trees
(bagging_boosting.ipynb:136, ``ntree_limit``) - with LightGBM's Python names.
"""
from __future__ import annotations

import ctypes
import weakref
import json
import os
import warnings
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import LightGBMError, c_str, check

__all__ = ["Dataset", "Booster", "LightGBMError", "params_to_string", "resolve_alias", "default_device"]


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
# ----------------------------------------------------------------------------
# Messages: printed by default, or handed to a logger registered with register_logger().
# ----------------------------------------------------------------------------
_logger = None
_info_method, _warning_method = "info", "warning"


def register_logger(logger, info_method_name: str = "info", warning_method_name: str = "warning") -> None:
    """Routes the package's messages (evaluation lines, early-stopping notes, warnings) to ``logger``, any
    object with the two named methods - e.g. a ``logging.Logger``.  ``None`` restores printing."""
    global _logger, _info_method, _warning_method
    if logger is not None:
        for name in (info_method_name, warning_method_name):
            if not callable(getattr(logger, name, None)):
                raise TypeError(f"Logger must provide '{name}' method")
    _logger, _info_method, _warning_method = logger, info_method_name, warning_method_name


def _log_info(msg: str) -> None:
    if _logger is not None:
        getattr(_logger, _info_method)(msg)
    else:
        print(msg)


def _log_warning(msg: str, stacklevel: int = 3) -> None:
    if _logger is not None:
        getattr(_logger, _warning_method)(msg)
    else:
        warnings.warn(msg, stacklevel=stacklevel + 1)


def _fmt_value(v: Any) -> str:
    if isinstance(v, bool) or isinstance(v, np.bool_):
        return "true" if v else "false"
    if isinstance(v, (list, tuple, set)):
        # a list of lists (interaction_constraints) keeps its brackets: [[0, 1], [2, 3]] -> "[0,1],[2,3]"
        return ",".join("[" + _fmt_value(x) + "]" if isinstance(x, (list, tuple, set)) else _fmt_value(x) for x in v)
    if isinstance(v, (float, np.floating)):
        return repr(float(v))
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if v is None:
        return ""
    return str(v)


def params_to_string(params: Optional[Dict[str, Any]]) -> str:
    """``{"a": 1, "b": [1, 2]}`` -> ``"a=1 b=1,2"``; ``None`` values are dropped."""
    if not params:
        return ""
    toks = []
    for k, v in params.items():
        if v is None or callable(v):
            continue
        s = _fmt_value(v)
        if s == "" or any(ch.isspace() for ch in s):
            if k in ("output_model", "input_model"):
                continue
            s = s.replace(" ", "")
            if s == "":
                continue
        toks.append(f"{k}={s}")
    return " ".join(toks)


_alias_cache: Optional[Dict[str, str]] = None


def _alias_map() -> Dict[str, str]:
    global _alias_cache
    if _alias_cache is None:
        lib, _ = _lib.load()
        table = json.loads(lib.lga_alias_table_json().decode())
        m = {}
        for canon, aliases in table.items():
            m[canon] = canon
            for a in aliases:
                m[a] = canon
        _alias_cache = m
    return _alias_cache


def resolve_alias(name: str) -> Optional[str]:
    """Canonical parameter name for ``name`` or ``None`` when unknown."""
    return _alias_map().get(name)


def pop_param(params: Dict[str, Any], canonical: str, default: Any = None) -> Any:
    """Removes every alias of ``canonical`` from ``params``; returns the value (canonical name wins)."""
    found = default
    seen = False
    for k in list(params.keys()):
        if resolve_alias(k) == canonical:
            v = params.pop(k)
            if k == canonical or not seen:
                found = v
            seen = True
    return found


def get_param(params: Dict[str, Any], canonical: str, default: Any = None) -> Any:
    for k, v in params.items():
        if k == canonical:
            return v
    for k, v in params.items():
        if resolve_alias(k) == canonical:
            return v
    return default


def default_device() -> str:
    """``"gpu"`` when the HIP core is loaded (a GPU is visible), else ``"cpu"``."""
    return "gpu" if _lib.is_hip() else "cpu"


def _with_device(params: Optional[Dict[str, Any]]) -> Dict[str, Any]:
    p = dict(params or {})
    if get_param(p, "device_type") is None:
        p["device_type"] = default_device()
    return p


def _warn_unknown(params: Dict[str, Any]):
    unknown = [k for k in params if resolve_alias(k) is None and not callable(params[k])]
    if unknown:
        _log_warning("Unknown parameter(s) ignored: " + ", ".join(sorted(unknown)), stacklevel=3)


_no_effect_warned: set = set()


def _warn_no_effect(params: Dict[str, Any]) -> None:
    """Parameters that are accepted but change results in upstream LightGBM and not here: say so, once per setting."""
    notes = []
    m = str(get_param(params, "monotone_constraints_method", "basic")).lower()
    if m in ("intermediate", "advanced"):
        notes.append(("monotone_constraints_method=" + m,
                      "monotone_constraints_method=%s is served by the basic method (same guarantee, more conservative splits)" % m))
    q = get_param(params, "num_grad_quant_bins", None)
    if q is not None and int(q) != 4:
        notes.append(("num_grad_quant_bins", "num_grad_quant_bins has no effect: quantised histograms always use 16 bits "
                                             "(use_quantized_grad=false selects full precision)"))
    if str(get_param(params, "reg_sqrt", False)).lower() in ("true", "1"):
        o = str(get_param(params, "objective", "regression")).lower()
        if o not in ("regression", "regression_l2", "l2", "mean_squared_error", "mse", "l2_root", "root_mean_squared_error", "rmse"):
            notes.append(("reg_sqrt+" + o, "reg_sqrt is honoured for objective=regression only; it has no effect with objective=%s" % o))
    for key, msg in notes:
        if key not in _no_effect_warned:
            _no_effect_warned.add(key)
            _log_warning(msg, stacklevel=4)


# ----------------------------------------------------------------------------
# data conversion
# ----------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class Sequence:
    """Rows served in batches (upstream LightGBM's ``lightgbm.Sequence``): subclass it with ``__len__`` and
    ``__getitem__`` - an int gives one row, a slice a batch of rows - and pass one object, or a list of them, as the
    ``data`` of a :class:`Dataset`.  The rows are read ``batch_size`` at a time into one matrix, which is then binned
    like any other (this build keeps the whole binned matrix in memory anyway: 1 byte per value)."""
    batch_size = 4096

    def __getitem__(self, idx):
        raise NotImplementedError("Sequence subclasses implement __getitem__ for ints and slices")

    def __len__(self):
        raise NotImplementedError("Sequence subclasses implement __len__")


def _read_sequences(seqs) -> np.ndarray:
    parts = []
    for seq in seqs:
        n, step = len(seq), max(int(getattr(seq, "batch_size", 4096) or 4096), 1)
        for lo in range(0, n, step):
            block = np.asarray(seq[lo:min(lo + step, n)], dtype=np.float64)
            parts.append(block.reshape(1, -1) if block.ndim == 1 else block)
    if not parts:
        raise ValueError("empty Sequence")
    return np.ascontiguousarray(np.vstack(parts))


def _to_matrix(data) -> Tuple[Any, List[str] | None]:
    """Returns (2-D float32/float64 C- or F-contiguous ndarray | cuda tensor, column names or None)."""
    names = None
    if isinstance(data, Sequence):
        data = _read_sequences([data])
    elif isinstance(data, (list, tuple)) and data and all(isinstance(d, Sequence) for d in data):
        data = _read_sequences(data)
    if _is_torch(data):
        import torch
        if data.dim() != 2:
            raise ValueError("data must be 2-dimensional")
        if data.is_cuda:
            if data.dtype not in (torch.float32, torch.float64):
                data = data.float()
            return data.contiguous(), None
        data = data.detach().numpy()
    if hasattr(data, "columns") and hasattr(data, "values"):   # pandas
        names = [str(c) for c in data.columns]
        data = data.values
    if hasattr(data, "toarray") and not isinstance(data, np.ndarray):  # scipy sparse
        data = data.toarray()
    arr = np.asarray(data)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    if arr.ndim != 2:
        raise ValueError("data must be 2-dimensional")
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    if not (arr.flags["C_CONTIGUOUS"] or arr.flags["F_CONTIGUOUS"]):
        arr = np.ascontiguousarray(arr)
    return arr, names


def _encode_pandas_categories(data, known: Optional[List[List[Any]]] = None):
    """pandas frames: columns of dtype category become their integer codes (NaN for missing / unknown levels).
    Returns (data, indices of those columns, their level lists); `known` pins the levels (validation and
    prediction frames are coded like the training frame)."""
    if not (hasattr(data, "columns") and hasattr(data, "dtypes")):
        return data, [], known
    cat_cols = [i for i, dt in enumerate(data.dtypes) if str(dt) == "category"]
    if not cat_cols:
        return data, [], known
    import pandas as pd
    out = data.copy()
    levels = []
    for j, i in enumerate(cat_cols):
        col = out.iloc[:, i]
        if known is not None and j < len(known):
            col = col.cat.set_categories(known[j])
        levels.append(list(col.cat.categories))
        codes = col.cat.codes.to_numpy().astype(np.float64)
        codes[codes < 0] = np.nan
        out[out.columns[i]] = pd.Series(codes, index=out.index)
    return out, cat_cols, levels


def _to_1d(x, dtype) -> np.ndarray:
    if _is_torch(x):
        x = x.detach().cpu().numpy()
    if hasattr(x, "values") and not isinstance(x, np.ndarray):
        x = x.values
    return np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=dtype)


def _mat_args(arr: np.ndarray):
    dtype = 0 if arr.dtype == np.float32 else 1
    row_major = 1 if arr.flags["C_CONTIGUOUS"] else 0
    return arr.ctypes.data_as(ctypes.c_void_p), dtype, arr.shape[0], arr.shape[1], row_major


# ----------------------------------------------------------------------------
# Native handles and interpreter shutdown.  Objects still alive when the interpreter exits (module-level
# boosters and datasets of a script) are released by an exit hook while the runtime underneath - the HIP
# runtime, torch's allocator and streams - is still whole; destructors that run later, during module
# teardown, leave the native side alone.
# ----------------------------------------------------------------------------
_live_objects: "weakref.WeakSet" = weakref.WeakSet()
_shutting_down = False
_exit_hook_installed = False


def _release_all_at_exit():
    global _shutting_down
    objs = list(_live_objects)
    for o in sorted(objs, key=lambda o: isinstance(o, Dataset)):     # boosters first: they hold their datasets
        try:
            o._free()
        except Exception:
            pass
    _shutting_down = True


def _track(obj):
    """Called when an object takes ownership of a native handle.  The hook is registered then - after torch
    has been imported on the GPU path - so that it runs before torch's own exit handlers."""
    global _exit_hook_installed
    _live_objects.add(obj)
    if not _exit_hook_installed:
        import atexit
        atexit.register(_release_all_at_exit)
        _exit_hook_installed = True


# ----------------------------------------------------------------------------
# Dataset
# ----------------------------------------------------------------------------
class Dataset:
    """Binned training / validation data.

    Construction is lazy like upstream: the matrix is binned at the first
    ``construct()`` (called by ``train``/``cv``/``Booster``).  ``reference``
    makes validation data share the training data's bin boundaries;
    ``subset`` creates zero-copy row subsets (CV folds).
    """

    def __init__(self, data, label=None, reference: Optional["Dataset"] = None, weight=None, init_score=None,
                 feature_name: Union[str, Sequence[str]] = "auto", params: Optional[Dict[str, Any]] = None,
                 free_raw_data: bool = True, bin_sample=None,
                 categorical_feature: Union[str, Sequence[Union[int, str]]] = "auto", group=None):
        self.data = data
        # ranking: sizes of the queries, in row order (the rows of one query are contiguous)
        self.group = group
        # columns holding category codes (non-negative integers): indices or names; "auto" takes the pandas
        # columns of dtype category (their codes are used).  A split on such a column sends one category left.
        self.categorical_feature = categorical_feature
        self.pandas_categorical: Optional[List[List[Any]]] = None
        self._has_categorical = False
        # rows to derive the bin boundaries from instead of this matrix (data-parallel ranks pass the
        # same gathered sample so that every shard is binned identically)
        self.bin_sample = bin_sample
        self.label = label
        self.reference = reference
        self.weight = weight
        self.init_score = init_score
        self.feature_name = feature_name
        self.params = dict(params or {})
        self.free_raw_data = free_raw_data
        self._handle: Optional[ctypes.c_void_p] = None
        self._parent: Optional["Dataset"] = None
        self._used_indices: Optional[np.ndarray] = None
        self._predictor = None

    # -- construction --
    def construct(self) -> "Dataset":
        if self._handle is not None:
            return self
        lib, hip = _lib.load()
        if self._parent is not None:
            self._parent.construct()
            idx = np.ascontiguousarray(self._used_indices, dtype=np.int32)
            h = ctypes.c_void_p()
            check(lib.lga_dataset_subset(self._parent._handle, idx.ctypes.data_as(ctypes.c_void_p), len(idx),
                                         ctypes.byref(h)))
            self._handle = h
            _track(self)
            self._has_categorical = self._parent._has_categorical
            self.pandas_categorical = self._parent.pandas_categorical
            self._set_fields()
            return self
        if self.data is None:
            raise LightGBMError("Cannot construct a Dataset whose raw data was freed")
        if isinstance(self.data, (str, os.PathLike)):
            path = os.fspath(self.data)
            with open(path, "rb") as f:
                is_cache = f.read(6) == b"LGABIN"
            if is_cache:
                # a binary cache written by save_binary(): boundaries, bins and fields come from the file
                h = ctypes.c_void_p()
                check(lib.lga_dataset_load_binary(c_str(path), ctypes.byref(h)))
                self._handle = h
                _track(self)
                self._set_fields()
                return self
            # a text file (CSV / TSV / LibSVM) with the application's conventions: label_column / header
            # parameters, <file>.weight / <file>.query / <file>.init side files
            from .cli import _load_table, _side_file
            label_column = get_param(self.params, "label_column", 0)
            if isinstance(label_column, str) and label_column.startswith("name:"):
                raise LightGBMError("label_column=name:... is not supported; give the column index")
            ref_cols = self.reference.construct().num_feature() if self.reference is not None else 0
            X, y = _load_table(path, int(label_column), str(get_param(self.params, "header", False)).lower() in ("true", "1"),
                               ref_cols)
            if ref_cols and X.shape[1] < ref_cols:
                X = np.hstack([X, np.zeros((X.shape[0], ref_cols - X.shape[1]))])
            self.data = X
            if self.label is None:
                self.label = y
            for attr, suffix in (("weight", ".weight"), ("group", ".query"), ("init_score", ".init")):
                if getattr(self, attr) is None:
                    setattr(self, attr, _side_file(path, suffix))
        if self.reference is not None:
            self.reference.construct()   # bin boundaries come from the reference
        raw, auto_cat, self.pandas_categorical = _encode_pandas_categories(
            self.data, self.reference.pandas_categorical if self.reference is not None else None)
        mat, names = _to_matrix(raw)
        params = _with_device(self.params)
        cat_idx = self._categorical_indices(names, auto_cat, mat.shape[1], params)
        if self.reference is not None:
            self._has_categorical = self.reference._has_categorical
        else:
            # host binning: category lookup, zeros that count as missing values, per-feature bin counts
            self._has_categorical = (bool(cat_idx) or str(get_param(params, "zero_as_missing", False)).lower() in ("true", "1")
                                     or get_param(params, "max_bin_by_feature", None) not in (None, "", [])
                                     or get_param(params, "forcedbins_filename", None) not in (None, ""))
            if cat_idx:
                params["categorical_feature"] = cat_idx
            else:
                pop_param(params, "categorical_feature")
        if self._has_categorical and _is_torch(mat):
            mat = mat.cpu().numpy()      # category lookup runs on the host; the bins are uploaded afterwards
        _warn_unknown(params)
        pstr = params_to_string(params)
        on_gpu = hip and get_param(params, "device_type") in ("gpu", "hip", "cuda", "rocm")
        if _is_torch(mat) and not on_gpu:
            mat = mat.cpu().numpy()
        nrow, ncol = int(mat.shape[0]), int(mat.shape[1])
        h = ctypes.c_void_p()
        if on_gpu:
            from . import _device
            _device.sync_stream(lib)
        # 1. bin boundaries: from the reference, from an explicit sample, or from this matrix
        if self.reference is not None:
            check(lib.lga_dataset_from_reference(self.reference._root_handle(), nrow, ctypes.byref(h)))
        elif self.bin_sample is not None:
            smp = np.ascontiguousarray(np.asarray(self.bin_sample, dtype=np.float64))
            if smp.ndim != 2 or smp.shape[1] != ncol:
                raise LightGBMError("bin_sample must be a 2-D array with the data's number of columns")
            check(lib.lga_dataset_from_sample(smp.ctypes.data_as(ctypes.c_void_p), smp.shape[0], ncol, nrow,
                                              c_str(pstr), ctypes.byref(h)))
        elif _is_torch(mat) and not self._has_categorical:
            dtype = 0 if str(mat.dtype).endswith("float32") else 1
            check(lib.lga_hip_dataset_from_device_mat(ctypes.c_void_p(mat.data_ptr()), dtype, nrow, ncol, c_str(pstr),
                                                      None, ctypes.byref(h)))
            mat_binned = True
        elif on_gpu and bool(get_param(params, "device_bin_find", True)) and not self._has_categorical:
            # host matrix, device dataset: boundaries are found and bins are built on the GPU
            ptr, dtype, _, _, rm = _mat_args(mat)
            check(lib.lga_hip_dataset_from_host_mat(ptr, dtype, nrow, ncol, rm, c_str(pstr), ctypes.byref(h)))
            mat_binned = True
        else:
            ptr, dtype, _, _, rm = _mat_args(mat)
            host_bins = (not on_gpu) or self._has_categorical
            check(lib.lga_dataset_from_mat(ptr, dtype, nrow, ncol, rm, c_str(pstr), None, 1 if host_bins else 0,
                                           ctypes.byref(h)))
            mat_binned = host_bins
        # 2. bins
        if not locals().get("mat_binned", False):
            if _is_torch(mat):
                dtype = 0 if str(mat.dtype).endswith("float32") else 1
                check(lib.lga_hip_dataset_bin_device(h, ctypes.c_void_p(mat.data_ptr()), dtype, nrow, ncol))
            else:
                ptr, dtype, _, _, rm = _mat_args(mat)
                if on_gpu and not self._has_categorical:
                    check(lib.lga_hip_dataset_upload_raw(h, ptr, dtype, nrow, ncol, rm))
                else:
                    check(lib.lga_dataset_bin_matrix(h, ptr, dtype, nrow, ncol, rm))
        if on_gpu:
            import torch
            torch.cuda.synchronize()
        self._handle = h
        _track(self)
        self._num_total_features = mat.shape[1]
        if isinstance(self.feature_name, (list, tuple)):
            names = [str(n) for n in self.feature_name]
        if names is not None and self.reference is None:
            if len(names) != mat.shape[1]:
                raise LightGBMError("Length of feature_name does not match the number of columns")
            clean = [n.replace(" ", "_") for n in names]
            check(lib.lga_dataset_set_feature_names(self._handle, c_str("\n".join(clean))))
        self._set_fields()
        if self.free_raw_data:
            self.data = None
        return self

    def _categorical_indices(self, names, auto_cat, ncol: int, params) -> List[int]:
        spec = self.categorical_feature
        if isinstance(spec, str) and spec == "auto":
            from_params = get_param(params, "categorical_feature", None)
            if from_params not in (None, "", []):
                spec = [int(v) for v in str(from_params).split(",")] if isinstance(from_params, str) else list(from_params)
            else:
                return sorted(auto_cat)
        out = set(auto_cat)
        for c in spec or []:
            if isinstance(c, str):
                pool = names if names is not None else (list(self.feature_name) if isinstance(self.feature_name, (list, tuple)) else None)
                if pool is None or c not in pool:
                    raise LightGBMError(f"categorical_feature: unknown column name {c!r}")
                out.add(pool.index(c))
            else:
                if not 0 <= int(c) < ncol:
                    raise LightGBMError(f"categorical_feature: column index {c} out of range")
                out.add(int(c))
        return sorted(out)

    def save_binary(self, filename: str) -> "Dataset":
        """Writes the binned dataset (bin boundaries, bins, label / weight / init_score) to ``filename``;
        ``Dataset(filename)`` loads it back without touching the raw data again."""
        self.construct()
        lib, _ = _lib.load()
        check(lib.lga_dataset_save_binary(self._root_handle(), c_str(os.fspath(filename))))
        return self

    def _root_handle(self):
        self.construct()
        return self._handle

    def _set_fields(self):
        if self.label is not None:
            self.set_label(self.label)
        if self.weight is not None:
            self.set_weight(self.weight)
        if self.init_score is not None:
            self.set_init_score(self.init_score)
        if self.group is not None:
            self.set_group(self.group)

    def _set_field(self, name: str, values, dtype):
        lib, _ = _lib.load()
        if values is None:
            check(lib.lga_dataset_set_field(self._handle, c_str(name), None, 0, 1))
            return
        arr = _to_1d(values, dtype)
        check(lib.lga_dataset_set_field(self._handle, c_str(name), arr.ctypes.data_as(ctypes.c_void_p), len(arr),
                                        0 if dtype == np.float32 else 1))

    def set_label(self, label) -> "Dataset":
        self.label = label
        if self._handle is not None and label is not None:
            self._set_field("label", label, np.float32)
        return self

    def set_weight(self, weight) -> "Dataset":
        self.weight = weight
        if self._handle is not None:
            self._set_field("weight", weight, np.float32)
        return self

    def set_init_score(self, init_score) -> "Dataset":
        """One initial score per row; multiclass: ``[rows, num_class]`` (or the flat ``[class][row]`` layout)."""
        if init_score is not None and not _is_torch(init_score) and np.ndim(init_score) == 2:
            init_score = np.ascontiguousarray(np.asarray(init_score, dtype=np.float64).T).reshape(-1)
        self.init_score = init_score
        if self._handle is not None:
            self._set_field("init_score", init_score, np.float64)
        return self

    def set_group(self, group) -> "Dataset":
        """Query sizes of ranking data: ``group=[10, 20]`` makes rows 0-9 one query and rows 10-29 the next."""
        self.group = group
        if self._handle is not None:
            self._set_field("group", group, np.float64)
        return self

    def get_group(self) -> Optional[np.ndarray]:
        if self._handle is None and isinstance(self.data, (str, os.PathLike)):
            self.construct()       # fields of a file-backed dataset live in the file
        if self._handle is None:
            return None if self.group is None else _to_1d(self.group, np.int64)
        g = self._get_field("group")
        return None if g is None else g.astype(np.int64)

    def _get_field(self, name: str) -> Optional[np.ndarray]:
        self.construct()
        lib, _ = _lib.load()
        n = ctypes.c_int64()
        check(lib.lga_dataset_get_field(self._handle, c_str(name), None, 0, ctypes.byref(n)))
        if n.value == 0:
            return None
        out = np.empty(n.value, dtype=np.float64)
        check(lib.lga_dataset_get_field(self._handle, c_str(name), out.ctypes.data_as(ctypes.c_void_p), n.value,
                                        ctypes.byref(n)))
        return out

    def get_label(self) -> Optional[np.ndarray]:
        if self._handle is None and isinstance(self.data, (str, os.PathLike)):
            self.construct()       # fields of a file-backed dataset live in the file
        if self._handle is None:
            return None if self.label is None else _to_1d(self.label, np.float64)
        return self._get_field("label")

    def get_weight(self) -> Optional[np.ndarray]:
        if self._handle is None and isinstance(self.data, (str, os.PathLike)):
            self.construct()       # fields of a file-backed dataset live in the file
        if self._handle is None:
            return None if self.weight is None else _to_1d(self.weight, np.float64)
        return self._get_field("weight")

    def get_init_score(self) -> Optional[np.ndarray]:
        if self._handle is None and isinstance(self.data, (str, os.PathLike)):
            self.construct()       # fields of a file-backed dataset live in the file
        if self._handle is None:
            return None if self.init_score is None else _to_1d(self.init_score, np.float64)
        return self._get_field("init_score")

    def set_reference(self, reference: "Dataset") -> "Dataset":
        """Bin this data with ``reference``'s boundaries; before construction only."""
        if self._handle is not None:
            raise LightGBMError("Cannot set reference after the Dataset was constructed; create a new Dataset")
        self.reference = reference
        return self

    def set_categorical_feature(self, categorical_feature) -> "Dataset":
        if self._handle is not None and list(categorical_feature) != list(self.categorical_feature or []):
            raise LightGBMError("Cannot set categorical features after the Dataset was constructed; create a new Dataset")
        self.categorical_feature = categorical_feature
        return self

    def set_feature_name(self, feature_name) -> "Dataset":
        self.feature_name = feature_name
        if self._handle is not None and isinstance(feature_name, (list, tuple)):
            lib, _ = _lib.load()
            if len(feature_name) != self.num_feature():
                raise LightGBMError("Length of feature_name does not match the number of columns")
            check(lib.lga_dataset_set_feature_names(self._handle, c_str("\n".join(str(n).replace(" ", "_") for n in feature_name))))
        return self

    def get_params(self) -> Dict[str, Any]:
        return dict(self.params)

    def get_ref_chain(self, ref_limit: int = 100) -> set:
        """This dataset and the datasets it takes its bin boundaries from, transitively."""
        chain, head = set(), self
        while head is not None and len(chain) < ref_limit and head not in chain:
            chain.add(head)
            head = head.reference
        return chain

    def get_field(self, field_name: str) -> Optional[np.ndarray]:
        if field_name == "group":
            return self.get_group()
        return self._get_field(field_name)

    def set_field(self, field_name: str, data) -> "Dataset":
        setter = {"label": self.set_label, "weight": self.set_weight, "init_score": self.set_init_score,
                  "group": self.set_group}
        if field_name not in setter:
            raise LightGBMError(f"unknown field {field_name!r}")
        return setter[field_name](data)

    def feature_num_bin(self, feature: Union[int, str]) -> int:
        if isinstance(feature, str):
            feature = self.get_feature_name().index(feature)
        return int(self.bin_info(int(feature))["num_bin"])

    def add_features_from(self, other: "Dataset") -> "Dataset":
        """Appends ``other``'s columns.  Both datasets must still hold their raw data (``free_raw_data=False``);
        the combined matrix is binned again at the next ``construct()``."""
        if self.data is None or other.data is None:
            raise LightGBMError("add_features_from needs the raw data of both datasets (free_raw_data=False)")
        a, an = _to_matrix(_encode_pandas_categories(self.data, None)[0])
        b, bn = _to_matrix(_encode_pandas_categories(other.data, None)[0])
        if _is_torch(a):
            a = a.cpu().numpy()
        if _is_torch(b):
            b = b.cpu().numpy()
        if a.shape[0] != b.shape[0]:
            raise LightGBMError("Cannot add features from a Dataset with a different number of rows")
        if self._handle is not None:
            names = self.get_feature_name()
            onames = other.get_feature_name() if other._handle is not None else (bn or [f"D2_Column_{i}" for i in range(b.shape[1])])
            clash = set(names) & set(onames)
            self.feature_name = names + [f"D2_{n}" if n in clash else n for n in onames]
            lib, _ = _lib.load()
            lib.lga_dataset_free(self._handle)
            self._handle = None
        self.data = np.hstack([np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)])
        return self

    def get_data(self):
        if self.data is None:
            raise LightGBMError("Raw data was freed (free_raw_data=True)")
        return self.data

    # -- derived datasets --
    def create_valid(self, data, label=None, weight=None, init_score=None, params=None, group=None) -> "Dataset":
        return Dataset(data, label=label, reference=self, weight=weight, init_score=init_score, group=group,
                       params=params if params is not None else self.params, free_raw_data=self.free_raw_data)

    def subset(self, used_indices: Iterable[int], params: Optional[Dict[str, Any]] = None) -> "Dataset":
        """Row subset sharing this dataset's bins (no re-binning, no copy)."""
        ds = Dataset(None, params=params if params is not None else self.params, free_raw_data=self.free_raw_data)
        ds._parent = self
        ds._used_indices = np.sort(np.asarray(list(used_indices) if not isinstance(used_indices, np.ndarray)
                                              else used_indices, dtype=np.int64))
        ds.reference = self
        if self.group is not None:
            # the subset's queries: what is left of each of this dataset's queries, in order
            sizes = _to_1d(self.group, np.int64)
            qid = np.repeat(np.arange(len(sizes)), sizes)
            if len(qid) and len(ds._used_indices) and ds._used_indices[-1] >= len(qid):
                raise LightGBMError("subset: row index beyond the rows covered by group")
            kept = qid[ds._used_indices]
            ds.group = np.bincount(kept, minlength=len(sizes))[np.unique(kept)] if len(kept) else np.zeros(0, np.int64)
        return ds

    # -- shape / names --
    def num_data(self) -> int:
        self.construct()
        lib, _ = _lib.load()
        n = ctypes.c_int64()
        check(lib.lga_dataset_num_data(self._handle, ctypes.byref(n)))
        return n.value

    def num_feature(self) -> int:
        self.construct()
        lib, _ = _lib.load()
        n = ctypes.c_int()
        check(lib.lga_dataset_num_feature(self._handle, ctypes.byref(n)))
        return n.value

    def get_feature_name(self) -> List[str]:
        self.construct()
        lib, _ = _lib.load()
        need = ctypes.c_int64()
        check(lib.lga_dataset_get_feature_names(self._handle, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(lib.lga_dataset_get_feature_names(self._handle, buf, need.value, ctypes.byref(need)))
        return buf.value.decode().split("\n")

    def bin_info(self, feature: int) -> Dict[str, Any]:
        """Bin boundaries of one (real) feature: ``num_bin``, ``missing_bin``, ``upper_bounds``."""
        self.construct()
        lib, _ = _lib.load()
        nb, mb, nu = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        up = np.empty(257, dtype=np.float64)
        check(lib.lga_dataset_bin_info(self._handle, feature, ctypes.byref(nb), ctypes.byref(mb),
                                       up.ctypes.data_as(ctypes.c_void_p), 257, ctypes.byref(nu)))
        return {"num_bin": nb.value, "missing_bin": mb.value, "upper_bounds": up[:nu.value].copy()}

    def get_bins(self) -> np.ndarray:
        """Feature-major bin matrix ``[used_features, rows]`` of the root dataset (for tests / inspection)."""
        self.construct()
        lib, _ = _lib.load()
        nf = ctypes.c_int()
        check(lib.lga_dataset_num_used_feature(self._handle, ctypes.byref(nf)))
        root = self
        while root._parent is not None:
            root = root._parent
        n = root.num_data()
        out = np.empty((nf.value, n), dtype=np.uint8)
        check(lib.lga_dataset_get_bins(self._handle, out.ctypes.data_as(ctypes.c_void_p), None))
        return out

    def _free(self):
        if getattr(self, "_handle", None) is not None:
            lib, _ = _lib.load()
            h, self._handle = self._handle, None
            lib.lga_dataset_free(h)

    def __del__(self):
        try:
            if not _shutting_down:
                self._free()
        except Exception:
            pass


# ----------------------------------------------------------------------------
# Booster
# ----------------------------------------------------------------------------
class Booster:
    """A gradient-boosted (or random-forest) tree ensemble."""

    def __init__(self, params: Optional[Dict[str, Any]] = None, train_set: Optional[Dataset] = None,
                 model_file: Optional[str] = None, model_str: Optional[str] = None, comm=None):
        lib, hip = _lib.load()
        self._handle = ctypes.c_void_p()
        _track(self)
        self.params = dict(params or {})
        self.best_iteration = 0
        self.best_score: Dict[str, Dict[str, float]] = {}
        self.train_set = train_set
        self.valid_sets: List[Dataset] = []
        self.name_valid_sets: List[str] = []
        self._train_data_name = "training"
        self._comm = comm
        self._init_predictor: Optional[Booster] = None
        self.pandas_categorical: Optional[List[List[Any]]] = None
        if train_set is not None:
            if not isinstance(train_set, Dataset):
                raise TypeError("train_set must be a Dataset")
            p = _with_device(self.params)
            _warn_unknown(p)
            _warn_no_effect(p)
            # the dataset decides where its bins live; keep the booster on the same device
            train_set.params = {**p, **train_set.params} if train_set._handle is None else train_set.params
            train_set.construct()
            comm_h = comm.handle if comm is not None else None
            if hip:
                from . import _device
                _device.sync_stream(lib)
            check(lib.lga_booster_create(train_set._handle, c_str(params_to_string(p)), comm_h,
                                         ctypes.byref(self._handle)))
            self.params = p
            self.pandas_categorical = train_set.pandas_categorical
        elif model_file is not None:
            with open(model_file, "r") as f:
                self._load(f.read())
        elif model_str is not None:
            self._load(model_str)
        else:
            raise TypeError("Need at least one of train_set, model_file or model_str to create a Booster")
        self._eval_names: Optional[List[Tuple[str, bool]]] = None

    def _load(self, text: str):
        lib, _ = _lib.load()
        # the levels of pandas category columns travel in the model's last line, as in upstream files
        tail = text.rfind("pandas_categorical:")
        if tail >= 0:
            try:
                self.pandas_categorical = json.loads(text[tail + len("pandas_categorical:"):].strip() or "null")
            except ValueError:
                self.pandas_categorical = None
        check(lib.lga_booster_from_string(c_str(text), ctypes.byref(self._handle)))

    def _free(self):
        if getattr(self, "_handle", None):
            lib, _ = _lib.load()
            h, self._handle = self._handle, None
            lib.lga_booster_free(h)

    def __del__(self):
        try:
            if not _shutting_down:
                self._free()
        except Exception:
            pass

    # -- training --
    def add_valid(self, data: Dataset, name: str) -> "Booster":
        lib, _ = _lib.load()
        if data.reference is None and data._parent is None and data is not self.train_set:
            data.reference = self.train_set
        data.construct()
        check(lib.lga_booster_add_valid(self._handle, data._handle, c_str(name)))
        self.valid_sets.append(data)
        self.name_valid_sets.append(name)
        return self

    def update(self, train_set: Optional[Dataset] = None, fobj: Optional[Callable] = None) -> bool:
        """One boosting iteration.  Returns True when training cannot continue (no split found)."""
        lib, _ = _lib.load()
        fin = ctypes.c_int()
        if fobj is None:
            check(lib.lga_booster_update(self._handle, ctypes.byref(fin)))
        else:
            preds = self._inner_predict(-1, raw=True)
            grad, hess = fobj(preds, self.train_set)
            return self.__boost(grad, hess)
        return fin.value == 1

    @staticmethod
    def update_group(boosters: List["Booster"], n_iters: int, eval_valid: bool = True):
        """``n_iters`` boosting iterations of several boosters in lockstep, inside the native library; after every
        iteration the first validation set of every booster is evaluated (``eval_valid``).  On the HIP engine small
        boosters (at most 65536 training rows, pointwise objective and metrics, plain trees) run the whole chunk on
        the device - one workgroup per booster does gradients, bag, tree, leaf values, score updates and metric
        sums of every iteration with no host in between - and all boosters of the call share ONE kernel launch:
        the folds of a cross-validation, the rows of a parameter grid.  The loop ends early after an iteration in
        which no booster found anything to split.  Returns ``(values[iteration, booster, metric],
        no_split[iteration, booster], iterations_run)``."""
        lib, _ = _lib.load()
        n = len(boosters)
        if _lib.is_hip():
            from . import _device
            _device.sync_stream(lib)
        names = boosters[0]._metric_names() if eval_valid else []
        stride = max(1, len(names))
        handles = (ctypes.c_void_p * n)(*[b._handle for b in boosters])
        values = np.zeros((n_iters, n, stride), dtype=np.float64)
        finished = np.zeros((n_iters, n), dtype=np.int32)
        done = ctypes.c_int()
        check(lib.lga_booster_group_update(handles, n, int(n_iters), 0 if eval_valid else -2,
                                           values.ctypes.data_as(ctypes.c_void_p), stride,
                                           finished.ctypes.data_as(ctypes.c_void_p), ctypes.byref(done)))
        return values[: done.value, :, : len(names)], finished[: done.value].astype(bool), done.value

    def __boost(self, grad, hess) -> bool:
        lib, _ = _lib.load()
        K = self.num_model_per_iteration()
        if K > 1 and np.ndim(grad) == 2:
            # [row][class] (the shape predictions are handed out in) -> the core's [class][row]
            grad, hess = np.asarray(grad).T, np.asarray(hess).T
        g = _to_1d(np.ascontiguousarray(grad).reshape(-1), np.float32)
        h = _to_1d(np.ascontiguousarray(hess).reshape(-1), np.float32)
        if len(g) != len(h):
            raise ValueError("grad and hess have different lengths")
        fin = ctypes.c_int()
        check(lib.lga_booster_update_custom(self._handle, g.ctypes.data_as(ctypes.c_void_p),
                                            h.ctypes.data_as(ctypes.c_void_p), len(g), ctypes.byref(fin)))
        return fin.value == 1

    def boost(self, grad, hess) -> bool:
        return self.__boost(grad, hess)

    def rollback_one_iter(self) -> "Booster":
        lib, _ = _lib.load()
        check(lib.lga_booster_rollback(self._handle))
        return self

    def refit(self, data, label, decay_rate: float = 0.9, weight=None, **kwargs) -> "Booster":
        """A new booster with this model's tree structures and leaf values refitted to ``data``:
        ``leaf = decay_rate * old + (1 - decay_rate) * new``, ``new`` being the Newton step of the rows that
        reach the leaf (``lambda_l1`` / ``lambda_l2`` / ``max_delta_step`` may be given as keywords)."""
        lib, _ = _lib.load()
        raw, _, _ = _encode_pandas_categories(data, self.pandas_categorical)
        mat, _ = _to_matrix(raw)
        if _is_torch(mat):
            mat = mat.cpu().numpy()
        y = _to_1d(label, np.float32)
        if len(y) != mat.shape[0]:
            raise LightGBMError("refit: label length differs from the number of rows")
        w = None if weight is None else _to_1d(weight, np.float32)
        out = Booster(model_str=self.model_to_string(num_iteration=-1))
        ptr, dtype, nr, nc, rm = _mat_args(mat)
        check(lib.lga_booster_refit(out._handle, ptr, dtype, nr, nc, rm, y.ctypes.data_as(ctypes.c_void_p),
                                    None if w is None else w.ctypes.data_as(ctypes.c_void_p), float(decay_rate),
                                    c_str(params_to_string(kwargs))))
        return out

    def reset_parameter(self, params: Dict[str, Any]) -> "Booster":
        lib, _ = _lib.load()
        check(lib.lga_booster_reset_parameter(self._handle, c_str(params_to_string(params))))
        self.params.update(params)
        return self

    def current_iteration(self) -> int:
        return self.num_trees() // self.num_model_per_iteration()

    def num_trees(self) -> int:
        lib, _ = _lib.load()
        n = ctypes.c_int()
        check(lib.lga_booster_num_trees(self._handle, ctypes.byref(n)))
        return n.value

    def num_feature(self) -> int:
        lib, _ = _lib.load()
        n = ctypes.c_int()
        check(lib.lga_booster_num_feature(self._handle, ctypes.byref(n)))
        return n.value

    def grow_stats(self) -> Dict[str, int]:
        """Rounds / materialised / used splits of the last tree (batched best-first growth)."""
        lib, _ = _lib.load()
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.lga_booster_stats(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"rounds": a.value, "materialized_splits": b.value, "used_splits": c.value}

    def profile(self) -> Dict[str, Any]:
        lib, _ = _lib.load()
        out, n = ctypes.c_char_p(), ctypes.c_int64()
        check(lib.lga_booster_profile_json(self._handle, ctypes.byref(out), ctypes.byref(n)))
        return json.loads(ctypes.string_at(out, n.value).decode())

    # -- evaluation --
    def _metric_names(self) -> List[Tuple[str, bool]]:
        if self._eval_names is None:
            lib, _ = _lib.load()
            buf = ctypes.create_string_buffer(1024)
            check(lib.lga_booster_eval_names(self._handle, buf, 1024))
            s = buf.value.decode()
            self._eval_names = [(t.split(":")[0], t.split(":")[1] == "1") for t in s.split(",") if t]
        return self._eval_names

    def _inner_predict(self, which: int, raw: bool = False) -> np.ndarray:
        lib, _ = _lib.load()
        n = ctypes.c_int64()
        check(lib.lga_booster_inner_nrows(self._handle, which, ctypes.byref(n)))
        K = self.num_model_per_iteration()
        out = np.empty(n.value * K, dtype=np.float64)
        check(lib.lga_booster_inner_predict(self._handle, which, 1 if raw else 0, out.ctypes.data_as(ctypes.c_void_p)))
        # multiclass: the core's [class][row] layout becomes one row of num_class values per data row
        return out if K == 1 else np.ascontiguousarray(out.reshape(K, n.value).T)

    def get_gradients(self) -> Tuple[np.ndarray, np.ndarray]:
        """Gradients and hessians of the last iteration over the training rows (float32; multiclass:
        ``[class][row]`` flattened) - what the trees of that iteration were fitted to."""
        lib, _ = _lib.load()
        n = ctypes.c_int64()
        check(lib.lga_booster_inner_nrows(self._handle, -1, ctypes.byref(n)))
        K = self.num_model_per_iteration()
        g = np.empty(n.value * K, dtype=np.float32)
        h = np.empty(n.value * K, dtype=np.float32)
        check(lib.lga_booster_get_gradients(self._handle, g.ctypes.data_as(ctypes.c_void_p),
                                            h.ctypes.data_as(ctypes.c_void_p)))
        return g, h

    def _eval(self, which: int, name: str, feval=None) -> List[Tuple[str, str, float, bool]]:
        lib, _ = _lib.load()
        names = self._metric_names()
        res: List[Tuple[str, str, float, bool]] = []
        if names:
            vals = np.empty(len(names), dtype=np.float64)
            n = ctypes.c_int()
            check(lib.lga_booster_eval(self._handle, which, vals.ctypes.data_as(ctypes.c_void_p), len(names),
                                       ctypes.byref(n)))
            for (mn, hb), v in zip(names, vals[: n.value]):
                res.append((name, mn, float(v), hb))
        if feval is not None:
            fevals = feval if isinstance(feval, (list, tuple)) else [feval]
            data = self.train_set if which < 0 else self.valid_sets[which]
            preds = self._inner_predict(which, raw=False)
            for f in fevals:
                r = f(preds, data)
                rs = r if isinstance(r, list) else [r]
                for (en, ev, ehb) in rs:
                    res.append((name, en, float(ev), bool(ehb)))
        return res

    def eval_train(self, feval=None):
        return self._eval(-1, self._train_data_name, feval)

    def eval_valid(self, feval=None):
        out = []
        for i, n in enumerate(self.name_valid_sets):
            out.extend(self._eval(i, n, feval))
        return out

    # -- prediction --
    def predict(self, data, start_iteration: int = 0, num_iteration: Optional[int] = None, raw_score: bool = False,
                pred_leaf: bool = False, pred_contrib: bool = False, device: Optional[str] = None,
                **kwargs) -> np.ndarray:
        """Predicts a dense matrix.  ``num_iteration=k`` uses the first k trees (staged prediction).
        ``pred_leaf``: leaf index per (row, tree).  ``pred_contrib``: SHAP values, ``[n, n_features + 1]`` in
        raw-score units; the last column is the expected value and every row sums to the raw prediction."""
        lib, hip = _lib.load()
        if isinstance(data, Dataset):
            raise TypeError("Cannot predict a Dataset; pass the raw matrix")
        if num_iteration is None:
            num_iteration = self.best_iteration if self.best_iteration > 0 else -1
        if num_iteration is None or num_iteration <= 0:
            num_iteration = -1
        data, _, _ = _encode_pandas_categories(data, self.pandas_categorical)
        mat, _ = _to_matrix(data)
        early = kwargs.get("pred_early_stop", get_param(self.params, "pred_early_stop", False))
        if early and not pred_leaf and not pred_contrib:
            info = self.objective_info()
            kind = info["objective"].split(" ")[0]
            if kind in ("binary", "multiclass", "multiclassova") and not info["average_output"]:
                freq = int(kwargs.get("pred_early_stop_freq", get_param(self.params, "pred_early_stop_freq", 10)))
                margin = float(kwargs.get("pred_early_stop_margin", get_param(self.params, "pred_early_stop_margin", 10.0)))
                return self._predict_early_stop(mat, start_iteration, num_iteration, raw_score, device, kind,
                                                info["sigmoid"], max(freq, 1), margin)
        ptype = 3 if pred_contrib else (2 if pred_leaf else (1 if raw_score else 0))
        used = ctypes.c_int()
        check(lib.lga_booster_used_iterations(self._handle, start_iteration, num_iteration, ctypes.byref(used)))
        nrow = mat.shape[0]
        K = self.num_model_per_iteration()
        use_gpu = hip and not pred_leaf and not pred_contrib and (device == "gpu" or _is_torch(mat) or
                                             (device is None and nrow * max(used.value, 1) * K >= 2_000_000))
        if use_gpu:
            from . import _device
            return _device.predict(lib, self._handle, mat, start_iteration, num_iteration, ptype, K)
        if _is_torch(mat):
            mat = mat.cpu().numpy()
        width = K * ((mat.shape[1] + 1) if pred_contrib else (used.value if pred_leaf else 1))
        out = np.empty(nrow * width, dtype=np.float64)
        ptr, dtype, nr, nc, rm = _mat_args(mat)
        check(lib.lga_booster_predict(self._handle, ptr, dtype, nr, nc, rm, start_iteration, num_iteration, ptype,
                                      out.ctypes.data_as(ctypes.c_void_p)))
        if pred_leaf:
            return out.reshape(nrow, width).astype(np.int32)
        if pred_contrib or K > 1:
            return out.reshape(nrow, width)
        return out

    def _predict_early_stop(self, mat, start_iteration, num_iteration, raw_score, device, kind, sigmoid, freq, margin):
        """Classification with ``pred_early_stop``: the raw score is built up ``freq`` iterations at a time, and a row
        whose margin (binary: 2|score|; multiclass: best minus second-best class score) exceeds ``margin`` keeps
        the score it has - the trees that are left are not evaluated for it."""
        lib, _ = _lib.load()
        used = ctypes.c_int()
        check(lib.lga_booster_used_iterations(self._handle, start_iteration, num_iteration, ctypes.byref(used)))
        K = self.num_model_per_iteration()
        if _is_torch(mat):
            mat = mat.cpu().numpy()
        mat = np.asarray(mat)
        n = mat.shape[0]
        raw = np.zeros((n, K), dtype=np.float64)
        active = np.arange(n)
        it, end = start_iteration, start_iteration + used.value
        while it < end and active.size:
            step = min(freq, end - it)
            part = self.predict(mat[active], start_iteration=it, num_iteration=step, raw_score=True, device=device,
                                pred_early_stop=False)
            raw[active] += np.asarray(part, dtype=np.float64).reshape(active.size, K)
            it += step
            cur = raw[active]
            if K == 1:
                m = 2.0 * np.abs(cur[:, 0])
            else:
                top = np.partition(cur, K - 2, axis=1)
                m = top[:, K - 1] - top[:, K - 2]
            active = active[m <= margin]
        if not raw_score:
            if kind == "multiclass":
                e = np.exp(raw - raw.max(axis=1, keepdims=True))
                raw = e / e.sum(axis=1, keepdims=True)
            else:
                raw = 1.0 / (1.0 + np.exp(-sigmoid * raw))
        return raw[:, 0] if K == 1 else raw

    # -- model IO --
    def model_to_string(self, num_iteration: Optional[int] = None, start_iteration: int = 0,
                        importance_type: Optional[str] = None) -> str:
        """``importance_type`` ("split" / "gain") of the ``feature_importances`` block; default: the booster's
        ``saved_feature_importance_type`` parameter (0 = split counts)."""
        lib, _ = _lib.load()
        if importance_type is None:
            importance_type = "gain" if int(get_param(self.params, "saved_feature_importance_type", 0) or 0) == 1 else "split"
        if num_iteration is None:
            num_iteration = self.best_iteration if self.best_iteration > 0 else -1
        out, n = ctypes.c_char_p(), ctypes.c_int64()
        check(lib.lga_booster_save_string(self._handle, start_iteration, num_iteration or -1,
                                          0 if importance_type == "split" else 1, ctypes.byref(out), ctypes.byref(n)))
        text = ctypes.string_at(out, n.value).decode()
        if self.pandas_categorical:
            text = text.replace("pandas_categorical:null", "pandas_categorical:" + json.dumps(self.pandas_categorical, default=str))
        return text

    def save_model(self, filename: str, num_iteration: Optional[int] = None, start_iteration: int = 0,
                   importance_type: Optional[str] = None) -> "Booster":
        text = self.model_to_string(num_iteration, start_iteration, importance_type)
        tmp = f"{filename}.tmp.{os.getpid()}"
        with open(tmp, "w") as f:
            f.write(text)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, filename)
        return self

    def model_from_string(self, model_str: str) -> "Booster":
        lib, _ = _lib.load()
        if self._handle:
            lib.lga_booster_free(self._handle)
        self._handle = ctypes.c_void_p()
        _track(self)
        self._load(model_str)
        self.train_set = None
        self.valid_sets, self.name_valid_sets = [], []
        self._eval_names = None
        return self

    def dump_model(self, num_iteration: Optional[int] = None, start_iteration: int = 0) -> Dict[str, Any]:
        lib, _ = _lib.load()
        out, n = ctypes.c_char_p(), ctypes.c_int64()
        check(lib.lga_booster_dump_json(self._handle, start_iteration, num_iteration or -1, ctypes.byref(out),
                                        ctypes.byref(n)))
        return json.loads(ctypes.string_at(out, n.value).decode())

    def feature_importance(self, importance_type: str = "split", iteration: Optional[int] = None) -> np.ndarray:
        lib, _ = _lib.load()
        out = np.zeros(self.num_feature(), dtype=np.float64)
        check(lib.lga_booster_feature_importance(self._handle, iteration or -1, 0 if importance_type == "split" else 1,
                                                 out.ctypes.data_as(ctypes.c_void_p)))
        return out.astype(np.int32) if importance_type == "split" else out

    def feature_name(self) -> List[str]:
        return self.dump_model(num_iteration=1)["feature_names"] if self.num_trees() else \
            json.loads(json.dumps(self.dump_model()))["feature_names"]

    def objective_info(self) -> Dict[str, Any]:
        lib, _ = _lib.load()
        o, a, s = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_double()
        check(lib.lga_booster_info(self._handle, ctypes.byref(o), ctypes.byref(a), ctypes.byref(s)))
        return {"objective": o.value.decode(), "average_output": bool(a.value), "sigmoid": s.value}

    def num_model_per_iteration(self) -> int:
        """Trees per boosting iteration (``num_class`` for multiclass objectives, else 1)."""
        lib, _ = _lib.load()
        n = ctypes.c_int()
        check(lib.lga_booster_num_class(self._handle, ctypes.byref(n)))
        return n.value

    def get_leaf_output(self, tree_id: int, leaf_id: int) -> float:
        return float(self._tree_arrays()[tree_id]["leaf_value"][leaf_id])

    def set_leaf_output(self, tree_id: int, leaf_id: int, value: float) -> "Booster":
        """Overwrites one leaf value of the model (scores accumulated during training are left alone)."""
        lib, _ = _lib.load()
        check(lib.lga_booster_set_leaf_value(self._handle, int(tree_id), int(leaf_id), float(value)))
        return self

    def shuffle_models(self, start_iteration: int = 0, end_iteration: int = -1, seed: int = 0) -> "Booster":
        """Puts the iterations ``[start_iteration, end_iteration)`` into a (seeded) random order; predictions over
        all iterations do not change, staged ones (``num_iteration``) do."""
        lib, _ = _lib.load()
        check(lib.lga_booster_shuffle(self._handle, int(start_iteration), int(end_iteration), int(seed)))
        return self

    def set_train_data_name(self, name: str) -> "Booster":
        self._train_data_name = name
        return self

    def attr(self, key: str) -> Optional[str]:
        return getattr(self, "_attr", {}).get(key)

    def set_attr(self, **kwargs) -> "Booster":
        """String attributes kept with the Python object (``None`` deletes one)."""
        store = self.__dict__.setdefault("_attr", {})
        for k, v in kwargs.items():
            if v is None:
                store.pop(k, None)
            elif not isinstance(v, str):
                raise ValueError("Only string values are accepted")
            else:
                store[k] = v
        return self

    def set_network(self, *args, **kwargs) -> "Booster":
        raise LightGBMError("socket networks are not used here: pass comm= (lightgbm_amd.parallel) to train(), "
                            "one process per GPU under torch.distributed")

    def free_network(self) -> "Booster":
        return self

    def lower_bound(self) -> float:
        """Smallest raw score the model can produce (sum of the per-tree minimum leaf values)."""
        mins = [float(np.min(t["leaf_value"])) for t in self._tree_arrays()]
        return self._combine_bounds(mins)

    def upper_bound(self) -> float:
        maxs = [float(np.max(t["leaf_value"])) for t in self._tree_arrays()]
        return self._combine_bounds(maxs)

    def _combine_bounds(self, per_tree: List[float]) -> float:
        if not per_tree:
            return 0.0
        total = float(np.sum(per_tree))
        return total / len(per_tree) if self.objective_info()["average_output"] else total

    def _tree_arrays(self) -> List[Dict[str, np.ndarray]]:
        from .models.ensemble import TreeEnsemble
        ens = TreeEnsemble.from_string(self.model_to_string(num_iteration=-1))
        return [{"leaf_value": t.leaf_value} for t in ens.trees]

    def get_split_value_histogram(self, feature: Union[int, str], bins=None, xgboost_style: bool = False):
        """Histogram of the thresholds the model splits ``feature`` (index or name) at: ``(counts, bin_edges)`` like
        ``numpy.histogram``; ``xgboost_style=True`` gives a two-column array (``SplitValue``, ``Count``) of the
        non-empty bins instead.  ``bins=None`` takes one bin per distinct threshold (at most the feature's count)."""
        model = self.dump_model(num_iteration=-1)
        names = model["feature_names"]
        if isinstance(feature, str):
            if feature not in names:
                raise LightGBMError(f"unknown feature name {feature!r}")
            feature = names.index(feature)
        values: List[float] = []

        def walk(node):
            if "split_feature" in node:
                if node["split_feature"] == feature:
                    if node.get("decision_type") == "==":
                        raise LightGBMError("Cannot compute split value histogram for the categorical feature")
                    values.append(float(node["threshold"]))
                walk(node["left_child"])
                walk(node["right_child"])
        for t in model["tree_info"]:
            walk(t["tree_structure"])
        if bins is None or (isinstance(bins, int) and xgboost_style):
            n_unique = len(np.unique(values))
            bins = max(min(n_unique, bins) if bins is not None else n_unique, 1)
        hist, edges = np.histogram(values, bins=bins)
        if xgboost_style:
            ret = np.column_stack((edges[1:], hist))
            return ret[ret[:, 1] > 0]
        return hist, edges

    def trees_to_dataframe(self):
        """One row per node of every tree (pandas), with the python-package's column set: ``tree_index,
        node_depth, node_index, left_child, right_child, parent_index, split_feature, split_gain, threshold,
        decision_type, missing_direction, missing_type, value, weight, count``."""
        import pandas as pd
        model = self.dump_model(num_iteration=-1)
        names = model["feature_names"]
        rows: List[Dict[str, Any]] = []

        def node_name(tree_index, node):
            if "split_index" in node:
                return f"{tree_index}-S{node['split_index']}"
            return f"{tree_index}-L{node.get('leaf_index', 0)}"

        def walk(tree_index, node, depth, parent):
            is_split = "split_index" in node
            row = {"tree_index": tree_index, "node_depth": depth, "node_index": node_name(tree_index, node),
                   "left_child": None, "right_child": None, "parent_index": parent, "split_feature": None,
                   "split_gain": None, "threshold": None, "decision_type": None, "missing_direction": None,
                   "missing_type": None}
            if is_split:
                row.update(left_child=node_name(tree_index, node["left_child"]),
                           right_child=node_name(tree_index, node["right_child"]),
                           split_feature=names[node["split_feature"]], split_gain=node["split_gain"],
                           threshold=node["threshold"], decision_type=node["decision_type"],
                           missing_direction="left" if node["default_left"] else "right",
                           missing_type=node["missing_type"], value=node["internal_value"],
                           weight=node["internal_weight"], count=node["internal_count"])
            else:
                row.update(value=node["leaf_value"], weight=node.get("leaf_weight"), count=node.get("leaf_count"))
            rows.append(row)
            if is_split:
                walk(tree_index, node["left_child"], depth + 1, row["node_index"])
                walk(tree_index, node["right_child"], depth + 1, row["node_index"])

        for t in model["tree_info"]:
            walk(t["tree_index"], t["tree_structure"], 1, None)
        return pd.DataFrame(rows)

    def eval(self, data: Dataset, name: str, feval=None):
        """Evaluates the metrics on ``data`` (added as a validation set on first use)."""
        if data is self.train_set:
            return self._eval(-1, name, feval)
        for i, d in enumerate(self.valid_sets):
            if d is data:
                return self._eval(i, name, feval)
        self.add_valid(data, name)
        return self._eval(len(self.valid_sets) - 1, name, feval)

    # pickling / copying go through the text model (training state is not kept)
    def __getstate__(self):
        return {"model": self.model_to_string(num_iteration=-1), "best_iteration": self.best_iteration,
                "best_score": getattr(self, "best_score", {}), "params": dict(self.params)}

    def __setstate__(self, state):
        self.__init__(params=state.get("params"), model_str=state["model"])
        self.best_iteration = state.get("best_iteration", 0)
        self.best_score = state.get("best_score", {})

    def __copy__(self):
        return self.__deepcopy__(None)

    def __deepcopy__(self, memo):
        b = Booster(params=dict(self.params), model_str=self.model_to_string(num_iteration=-1))
        b.best_iteration = self.best_iteration
        return b

    def _prepend(self, first: "Booster"):
        lib, _ = _lib.load()
        check(lib.lga_booster_prepend(self._handle, first._handle))

    def free_dataset(self) -> "Booster":
        self.train_set = None
        self.valid_sets = []
        return self
