"""Multi-GPU training: data-parallel / feature-parallel learners (RCCL over xGMI) and task-parallel grid sharding."""
from .comm import DeviceLoopbackComm, RcclComm, ThreadComm, TorchComm, default_comm
from .data_parallel import distributed_dataset, gather_bin_sample, train_data_parallel

__all__ = ["DeviceLoopbackComm", "RcclComm", "TorchComm", "ThreadComm", "default_comm", "distributed_dataset", "gather_bin_sample",
           "train_data_parallel"]
