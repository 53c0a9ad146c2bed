"""Tree-parallel random forests: the trees of a bagged forest do not depend on each other
("bagging_boosting.ipynb":11 - "fitted in parallel"), so P ranks can each grow n/P of them on the full
data and the pieces are concatenated.  Iteration k draws the same bag and feature subset wherever it is
trained (``iteration_offset``), so the merged forest is the forest one process would have grown.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

from ..basic import Booster, Dataset
from ..engine import train


def tree_share(num_trees: int, world: int, rank: int) -> range:
    """Iterations rank ``rank`` of ``world`` grows (contiguous, sizes differ by at most one)."""
    return range(num_trees * rank // world, num_trees * (rank + 1) // world)


def train_forest_part(params: Dict[str, Any], train_set: Dataset, iterations: range) -> Optional[Booster]:
    """Grows iterations ``iterations`` of the forest described by ``params`` (boosting=rf)."""
    if len(iterations) == 0:
        return None
    p = {**params, "boosting": "rf", "iteration_offset": iterations.start, "tree_learner": "serial"}
    return train(p, train_set, num_boost_round=len(iterations))


def merge_forests(parts: Sequence[Optional[str]]) -> Booster:
    """Concatenates model strings (in order) into one averaged ensemble."""
    texts = [t for t in parts if t]
    if not texts:
        raise ValueError("no forest parts to merge")
    acc = Booster(model_str=texts[0])
    for t in texts[1:]:
        nxt = Booster(model_str=t)
        nxt._prepend(acc)
        acc = nxt
    return acc


def train_forest_tree_parallel(params: Dict[str, Any], X, y, num_trees: int, weight=None, group=None) -> Booster:
    """Every rank of the ``torch.distributed`` group holds the full data, grows its share of the trees on
    its own GPU and receives the whole forest."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ds = Dataset(X, label=y, weight=weight, params=params)
    part = train_forest_part(params, ds, tree_share(num_trees, world, rank))
    texts: List[Any] = [None] * world
    dist.all_gather_object(texts, part.model_to_string(num_iteration=-1) if part is not None else None, group=group)
    return merge_forests(texts)
