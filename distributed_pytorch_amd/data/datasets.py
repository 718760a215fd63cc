"""Datasets: ``ImageFolder`` (PIL) and the synthetic pinned-memory source.

``ImageFolder`` reproduces ``torchvision.datasets.ImageFolder`` as used at
reference ``nd_imagenet.py:208-232``: ``DIR/<class>/<img>``, classes sorted,
``(sample, class_index)``.  ``SyntheticImageNet`` is the headline data path of
BASELINE.json: uint8 224x224 batches staged in pinned host memory.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def pil_loader(path: str):
    from PIL import Image

    with open(path, "rb") as f:
        return Image.open(f).convert("RGB")


class ImageFolder(torch.utils.data.Dataset):
    def __init__(self, root: str, transform: Optional[Callable] = None,
                 loader: Callable = pil_loader):
        if not os.path.isdir(root):
            raise FileNotFoundError("dataset directory %r does not exist" % root)
        self.root, self.transform, self.loader = root, transform, loader
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError("no class folders in %r" % root)
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples: List[Tuple[str, int]] = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for name in sorted(files):
                    if name.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, name), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError("no images found under %r" % root)
        self.targets = [s[1] for s in self.samples]

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, index: int):
        path, target = self.samples[index]
        img = self.loader(path)
        if self.transform is not None:
            img = self.transform(img)
        return img, target


def fast_collate(batch, memory_format=torch.contiguous_format, nhwc: bool = False):
    """PIL images -> ONE uint8 tensor, no float conversion on the host
    (reference ``imagenet_ddp_apex.py:26-39``).  ``nhwc=True`` keeps PIL's native
    HWC order, which is what the GPU normalise kernel wants (no host transpose)."""
    imgs = [b[0] for b in batch]
    targets = torch.tensor([b[1] for b in batch], dtype=torch.int64)
    w, h = imgs[0].size
    if nhwc:
        out = torch.empty((len(imgs), h, w, 3), dtype=torch.uint8)
    else:
        out = torch.zeros((len(imgs), 3, h, w), dtype=torch.uint8).contiguous(
            memory_format=memory_format)
    for i, img in enumerate(imgs):
        arr = np.array(img, dtype=np.uint8)
        if arr.ndim < 3:
            arr = np.repeat(arr[:, :, None], 3, axis=2)
        if nhwc:
            out[i] = torch.from_numpy(np.ascontiguousarray(arr))
        else:
            out[i] += torch.from_numpy(np.ascontiguousarray(np.rollaxis(arr, 2)))
    return out, targets


class SyntheticImageNet:
    """Iterable of ``(uint8 [B,224,224,3] pinned, int64 [B] pinned)`` batches.

    A small ring of distinct pre-generated pinned batches is cycled so every
    step performs a real async H2D copy of fresh bytes; ``len()`` follows the
    sampler arithmetic of a real epoch (``ceil(N / world / batch)``).
    """

    def __init__(self, batch_size: int, num_samples: int = 1281167, world_size: int = 1,
                 rank: int = 0, image_size: int = 224, num_classes: int = 1000, ring: int = 4,
                 seed: int = 0, pin: Optional[bool] = None, steps: int = 0):
        self.batch_size, self.num_classes = batch_size, num_classes
        per_rank = -(-num_samples // world_size)
        self.num_batches = max(1, -(-per_rank // batch_size))
        if steps:
            self.num_batches = min(self.num_batches, steps)
        pin = torch.cuda.is_available() if pin is None else pin
        g = torch.Generator().manual_seed(seed * 1000003 + rank)
        self.images, self.targets = [], []
        for _ in range(ring):
            img = torch.randint(0, 256, (batch_size, image_size, image_size, 3), dtype=torch.uint8,
                                generator=g)
            tgt = torch.randint(0, num_classes, (batch_size,), dtype=torch.int64, generator=g)
            if pin:
                img, tgt = img.pin_memory(), tgt.pin_memory()
            self.images.append(img)
            self.targets.append(tgt)

    def __len__(self) -> int:
        return self.num_batches

    def __iter__(self):
        for i in range(self.num_batches):
            j = i % len(self.images)
            yield self.images[j], self.targets[j]
