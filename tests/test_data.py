"""Data pipeline: DistributedSampler parity with torch, collate, transforms, synthetic source, prefetcher."""
import os

import numpy as np
import pytest
import torch

from distributed_pytorch_amd.data import transforms as T
from distributed_pytorch_amd.data.datasets import ImageFolder, SyntheticImageNet, fast_collate
from distributed_pytorch_amd.data.prefetcher import StreamPrefetcher
from distributed_pytorch_amd.data.sampler import DistributedSampler
from distributed_pytorch_amd.ops import functional as F


@pytest.mark.parametrize("n,world", [(10, 4), (17, 3), (64, 8), (5, 8)])
@pytest.mark.parametrize("shuffle", [True, False])
def test_sampler_index_parity_with_torch(n, world, shuffle):
    from torch.utils.data.distributed import DistributedSampler as TorchSampler

    ds = list(range(n))
    for epoch in (0, 3):
        seen = []
        for rank in range(world):
            ours = DistributedSampler(ds, world, rank, shuffle=shuffle, seed=7)
            ref = TorchSampler(ds, world, rank, shuffle=shuffle, seed=7)
            ours.set_epoch(epoch)
            ref.set_epoch(epoch)
            assert list(ours) == list(ref) and len(ours) == len(ref)
            seen += list(ours)
        assert set(seen) == set(range(n))          # padded, complete cover


def _make_folder(root, classes=("a", "b"), per=3, size=(40, 30)):
    from PIL import Image

    rng = np.random.RandomState(0)
    for c in classes:
        os.makedirs(os.path.join(root, c))
        for i in range(per):
            arr = rng.randint(0, 255, (size[1], size[0], 3), dtype=np.uint8)
            if i == 0:
                Image.fromarray(arr[:, :, 0], mode="L").save(os.path.join(root, c, "g%d.png" % i))   # grayscale
            else:
                Image.fromarray(arr).save(os.path.join(root, c, "i%d.png" % i))


def test_imagefolder_transforms_and_fast_collate(tmp_path):
    _make_folder(str(tmp_path / "train"))
    ds = ImageFolder(str(tmp_path / "train"), T.Compose([T.RandomResizedCrop(24), T.RandomHorizontalFlip()]))
    assert len(ds) == 6 and ds.classes == ["a", "b"] and ds.class_to_idx == {"a": 0, "b": 1}
    batch = [ds[i] for i in range(6)]
    x, t = fast_collate(batch)                       # reference layout: uint8 NCHW
    assert x.dtype == torch.uint8 and x.shape == (6, 3, 24, 24) and t.tolist() == [0, 0, 0, 1, 1, 1]
    xh, _ = fast_collate(batch, nhwc=True)
    assert xh.shape == (6, 24, 24, 3) and torch.equal(xh.permute(0, 3, 1, 2), x)
    val = ImageFolder(str(tmp_path / "train"), T.Compose([T.Resize(32), T.CenterCrop(24), T.ToTensor(),
                                                          T.Normalize(F.IMAGENET_MEAN, F.IMAGENET_STD)]))
    img, _ = val[1]
    assert img.shape == (3, 24, 24) and img.dtype == torch.float32


def test_normalize_u8_matches_reference_prefetcher_math():
    u8 = torch.randint(0, 256, (2, 5, 6, 3), dtype=torch.uint8)
    mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255]).view(1, 3, 1, 1)
    std = torch.tensor([0.229 * 255, 0.224 * 255, 0.225 * 255]).view(1, 3, 1, 1)
    ref = u8.permute(0, 3, 1, 2).float().sub_(mean).div_(std)        # imagenet_ddp_apex.py:339-340
    ours = F.normalize_u8(u8, torch.float32)
    assert torch.allclose(ours.permute(0, 3, 1, 2), ref, atol=1e-6)
    ours2 = F.normalize_u8(u8.permute(0, 3, 1, 2).contiguous(), torch.float32, input_layout="nchw")
    assert torch.allclose(ours2, ours)


def test_synthetic_source_and_prefetcher_cpu():
    src = SyntheticImageNet(4, num_samples=40, world_size=2, rank=1, image_size=16, num_classes=10, steps=0, pin=False)
    assert len(src) == 5                               # ceil(ceil(40/2)/4)
    pre = StreamPrefetcher(src, torch.device("cpu"), dtype=torch.float32)
    n = 0
    for images, target in pre:
        assert images.shape == (4, 16, 16, 3) and images.dtype == torch.float32
        assert target.dtype == torch.int64 and int(target.max()) < 10
        n += 1
    assert n == 5
    assert pre.next() == (None, None)
