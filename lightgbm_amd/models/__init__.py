"""Model-level objects: the tree-ensemble file format as NumPy arrays (an independent reader of the text model)."""
from .ensemble import TreeArrays, TreeEnsemble

__all__ = ["TreeArrays", "TreeEnsemble"]
