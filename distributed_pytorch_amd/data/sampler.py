"""DistributedSampler — pure index math, no communication.

Index-for-index identical to ``torch.utils.data.distributed.DistributedSampler``
as used by the reference (``imagenet_ddp.py:178,202``): ``randperm`` seeded with
``seed + epoch``, padded by wrap-around to a multiple of the world size (or
truncated with ``drop_last``), then strided ``indices[rank::world]``.
"""
from __future__ import annotations

import math
from typing import Iterator, Optional

import torch


class DistributedSampler:
    def __init__(self, dataset_or_len, num_replicas: Optional[int] = None,
                 rank: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        if num_replicas is None or rank is None:
            from ..parallel import process_group as pg

            if not pg.is_initialized():
                raise RuntimeError("DistributedSampler needs num_replicas/rank or an initialised group")
            num_replicas = pg.get_world_size() if num_replicas is None else num_replicas
            rank = pg.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError("invalid rank {} for {} replicas".format(rank, num_replicas))
        self.n = dataset_or_len if isinstance(dataset_or_len, int) else len(dataset_or_len)
        self.num_replicas, self.rank = num_replicas, rank
        self.shuffle, self.seed, self.drop_last = shuffle, seed, drop_last
        self.epoch = 0
        if drop_last and self.n % num_replicas != 0:
            self.num_samples = math.ceil((self.n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(self.n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(self.n, generator=g).tolist()
        else:
            indices = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad <= len(indices):
                indices += indices[:pad]
            else:
                indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[: self.total_size]
        return iter(indices[self.rank: self.total_size: self.num_replicas])
