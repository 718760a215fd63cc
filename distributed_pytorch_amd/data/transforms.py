"""PIL/numpy image transforms (torchvision is not installed in the target image).

Semantics follow the torchvision transforms the reference composes
(``nd_imagenet.py:208-232``): ``RandomResizedCrop(224)`` (scale 0.08-1, ratio
3/4-4/3, 10 tries then centre fallback, bilinear), ``RandomHorizontalFlip``,
``Resize(256)`` (shorter side), ``CenterCrop(224)``, ``ToTensor``, ``Normalize``.
"""
from __future__ import annotations

import math
import random
from typing import Sequence

import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class RandomResizedCrop:
    def __init__(self, size: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size, self.scale, self.ratio = (size, size), scale, ratio

    def get_params(self, width: int, height: int):
        area = width * height
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * random.uniform(*self.scale)
            aspect = math.exp(random.uniform(*log_ratio))
            w = int(round(math.sqrt(target * aspect)))
            h = int(round(math.sqrt(target / aspect)))
            if 0 < w <= width and 0 < h <= height:
                return random.randint(0, height - h), random.randint(0, width - w), h, w
        in_ratio = width / height
        if in_ratio < self.ratio[0]:
            w, h = width, int(round(width / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            h, w = height, int(round(height * self.ratio[1]))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, img):
        from PIL import Image

        top, left, h, w = self.get_params(*img.size)
        return img.resize(self.size, Image.BILINEAR, box=(left, top, left + w, top + h))


class RandomHorizontalFlip:
    def __init__(self, p: float = 0.5):
        self.p = p

    def __call__(self, img):
        from PIL import Image

        return img.transpose(Image.FLIP_LEFT_RIGHT) if random.random() < self.p else img


class Resize:
    def __init__(self, size: int):
        self.size = size

    def __call__(self, img):
        from PIL import Image

        w, h = img.size
        if w <= h:
            nw, nh = self.size, int(self.size * h / w)
        else:
            nw, nh = int(self.size * w / h), self.size
        return img.resize((nw, nh), Image.BILINEAR)


class CenterCrop:
    def __init__(self, size: int):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
        return img.crop((left, top, left + self.size, top + self.size))


class ToTensor:
    """PIL -> float32 CHW in [0,1]."""

    def __call__(self, img):
        arr = np.asarray(img, dtype=np.uint8)
        if arr.ndim < 3:
            arr = arr[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).float().div_(255.0)


class Normalize:
    def __init__(self, mean: Sequence[float], std: Sequence[float]):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std
