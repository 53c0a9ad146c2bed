"""lightgbm_amd: an MI355X-native gradient-boosted decision tree framework.

LightGBM-style leaf-wise histogram boosting and random-forest mode with
``Dataset`` / ``train`` / ``cv`` / ``predict`` entry points, early-stopped k-fold
CV, a resumable parameter-grid CV search, and the LightGBM text model format.
The compute path is hand-written HIP for gfx950 (``csrc/hip``) driven by a C++
core; a same-numerics OpenMP engine serves CPU runs and as the test oracle.
"""
import os as _os

# The host driver of some machines shares device memory between processes through dmabuf only; RCCL (and CUDA
# tensor sharing between processes) then needs this before the HIP runtime starts - which is at the first GPU
# call, not at import - so setting it here covers every launcher that imports the package first.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .basic import Booster, Dataset, LightGBMError, Sequence, default_device, params_to_string, register_logger, resolve_alias
from .callback import EarlyStopException, early_stopping, log_evaluation, print_evaluation, record_evaluation, reset_parameter
from .engine import CVBooster, cv, make_folds, train

__version__ = "0.1.0"

_LAZY = {
    "LGBMModel": "sklearn", "LGBMRegressor": "sklearn", "LGBMClassifier": "sklearn", "LGBMRanker": "sklearn",
    "LGBMRandomForestRegressor": "sklearn",
    "plot_importance": "plotting", "plot_metric": "plotting", "plot_split_value_histogram": "plotting",
    "plot_tree": "plotting", "create_tree_digraph": "plotting",
}


def __getattr__(name):
    # scikit-learn / matplotlib are only imported when one of these names is asked for
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

__all__ = [
    "Dataset", "Booster", "CVBooster", "LightGBMError", "Sequence", "train", "cv", "make_folds",
    "early_stopping", "log_evaluation", "print_evaluation", "record_evaluation", "reset_parameter",
    "EarlyStopException", "default_device", "params_to_string", "resolve_alias", "register_logger",
]
