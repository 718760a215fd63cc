from .distributed import DistributedDataParallel, GradArena
from .sync_batchnorm import convert_syncbn_model
from .comm import reduce_tensor

__all__ = ["DistributedDataParallel", "GradArena", "convert_syncbn_model", "reduce_tensor"]
