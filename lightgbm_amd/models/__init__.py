"""Model-level objects: the tree-ensemble file format as NumPy arrays, and the two model families."""
from .ensemble import TreeArrays, TreeEnsemble
from .families import GBDT, RandomForest

__all__ = ["TreeArrays", "TreeEnsemble", "GBDT", "RandomForest"]
