"""Multi-GPU training: data-parallel / feature-parallel learners (RCCL over xGMI), tree-parallel forests and
task-parallel grid sharding."""
from .comm import DeviceLoopbackComm, RcclComm, ThreadComm, TorchComm, default_comm
from .data_parallel import distributed_dataset, gather_bin_sample, train_data_parallel
from .forest import merge_forests, train_forest_part, train_forest_tree_parallel, tree_share

__all__ = ["DeviceLoopbackComm", "RcclComm", "TorchComm", "ThreadComm", "default_comm", "distributed_dataset", "gather_bin_sample",
           "train_data_parallel", "train_forest_tree_parallel", "train_forest_part", "merge_forests", "tree_share"]
