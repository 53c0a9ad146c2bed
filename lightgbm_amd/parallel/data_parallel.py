"""Data-parallel training helpers: rows are sharded over ranks (one process per GPU).

Every rank bins its shard with the SAME boundaries: each rank contributes
``bin_construct_sample_cnt / world_size`` sampled rows, the samples are
all-gathered once, and every rank derives the bin mappers from the identical
gathered sample (no broadcast of boundaries needed).  Training then runs the
ordinary ``train`` loop with ``tree_learner="data"`` and a communicator; per
growth round one integer all-reduce merges the new leaf histograms.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from ..basic import Dataset, get_param
from .comm import default_comm


def gather_bin_sample(X_local, params: Optional[Dict[str, Any]] = None, group=None, seed: int = 1) -> np.ndarray:
    """All ranks return the same float64 sample [k, F] drawn evenly from every rank's shard."""
    import torch
    import torch.distributed as dist
    P = dist.get_world_size(group)
    rank = dist.get_rank(group)
    want = int(get_param(params or {}, "bin_construct_sample_cnt", 200000))
    per_rank = (want + P - 1) // P
    n = int(X_local.shape[0])
    F = int(X_local.shape[1])
    k = min(per_rank, n)
    rng = np.random.RandomState(seed + 7919 * rank)
    idx = np.sort(rng.choice(n, size=k, replace=False)) if k < n else np.arange(n)
    if hasattr(X_local, "data_ptr"):
        rows = X_local[torch.as_tensor(idx, device=X_local.device)].double().cpu()
    else:
        rows = torch.from_numpy(np.asarray(X_local)[idx].astype(np.float64))
    buf = torch.full((per_rank, F), float("nan"), dtype=torch.float64)
    buf[:k] = rows
    cnt = torch.tensor([k], dtype=torch.int64)
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu:
        buf, cnt = buf.cuda(), cnt.cuda()
    bufs = [torch.empty_like(buf) for _ in range(P)]
    cnts = [torch.empty_like(cnt) for _ in range(P)]
    dist.all_gather(bufs, buf, group=group)
    dist.all_gather(cnts, cnt, group=group)
    parts = [b[: int(c.item())].cpu().numpy() for b, c in zip(bufs, cnts)]
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def distributed_dataset(X_local, label=None, weight=None, params: Optional[Dict[str, Any]] = None, group=None,
                        **kwargs) -> Dataset:
    """This rank's shard as a Dataset binned with globally agreed boundaries."""
    sample = gather_bin_sample(X_local, params, group)
    return Dataset(X_local, label=label, weight=weight, params=params, bin_sample=sample, **kwargs)


def train_data_parallel(params: Dict[str, Any], X_local, y_local, num_boost_round: int = 100, group=None,
                        weight=None, tree_learner: str = "data", **train_kwargs):
    """Trains one model over all ranks' shards; every rank returns the same booster.

    ``tree_learner="voting"`` sums only the histogram rows of the features the ranks vote for (``top_k`` votes
    per rank and leaf; CPU engine - the device engine sums whole histograms by reduce-scatter either way)."""
    from ..engine import train
    if tree_learner not in ("data", "voting"):
        raise ValueError("rows are sharded by tree_learner='data' and 'voting'")
    p = dict(params)
    p["tree_learner"] = tree_learner
    ds = distributed_dataset(X_local, y_local, weight=weight, params=p, group=group)
    comm = default_comm(group)
    try:
        return train(p, ds, num_boost_round=num_boost_round, comm=comm, **train_kwargs)
    finally:
        if comm is not None:
            comm.close()
