"""Model zoo registry: the ``--arch`` choices (reference ``model_names`` at
``nd_imagenet.py:22-24`` lists every lowercase callable of ``torchvision.models``;
torchvision is not a dependency here, the zoo is native NHWC)."""
from __future__ import annotations

from typing import Callable, Dict, List

from . import alexnet_vgg as _av
from . import convnext as _cx
from . import densenet as _dn
from . import efficientnet as _ef
from . import googlenet as _gn
from . import maxvit as _mx
from . import inception as _in
from . import mobilenet as _mb
from . import regnet as _rg
from . import resnet as _rn
from . import shufflenet as _sh
from . import squeezenet as _sq
from . import swin as _sw
from . import vit as _vt

_REGISTRY: Dict[str, Callable] = {}
for _mod in (_rn, _av, _sq, _dn, _mb, _gn, _rg, _ef, _in, _cx, _sh, _vt, _sw, _mx):
    for _name in dir(_mod):
        _fn = getattr(_mod, _name)
        if _name.islower() and not _name.startswith("_") and callable(_fn) \
                and getattr(_fn, "__module__", "") == _mod.__name__ \
                and _name not in ("conv3x3", "conv1x1", "run_fused_sequential", "block_params"):
            _REGISTRY[_name] = _fn


def list_models() -> List[str]:
    return sorted(_REGISTRY)


def create_model(arch: str, pretrained: bool = False, pretrained_path: str = "", **kw):
    """``models.__dict__[arch](pretrained=...)`` equivalent.  There is no network
    on the target machines, so ``pretrained`` loads a local torchvision-format
    state_dict given by ``pretrained_path`` / ``$DPA_PRETRAINED_DIR/<arch>.pth``."""
    if arch not in _REGISTRY:
        raise KeyError("unknown arch %r; choices: %s" % (arch, ", ".join(list_models())))
    model = _REGISTRY[arch](**kw)
    if pretrained:
        import os

        import torch

        from ..checkpoint import load_model_state

        path = pretrained_path or os.path.join(os.environ.get("DPA_PRETRAINED_DIR", "."),
                                               arch + ".pth")
        if not os.path.isfile(path):
            raise FileNotFoundError(
                "--pretrained needs a local weights file (no network): %s not found" % path)
        sd = torch.load(path, map_location="cpu", weights_only=True)
        load_model_state(model, sd.get("state_dict", sd))
    return model


__dict__ = _REGISTRY  # noqa: A001  (mirrors ``models.__dict__[arch]`` usage)
