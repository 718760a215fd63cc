from .datasets import ImageFolder, SyntheticImageNet, fast_collate
from .prefetcher import StreamPrefetcher
from .sampler import DistributedSampler

__all__ = ["ImageFolder", "SyntheticImageNet", "fast_collate", "StreamPrefetcher",
           "DistributedSampler"]
