"""distributed_pytorch_amd — MI355X-native data-parallel ImageNet training framework.

Capability parity target: the Esthesia/distributed-pytorch scripts (ND / DP / DDP /
Apex-AMP trainers).  PyTorch-ROCm is the tensor/autograd layer only: hot-path ops
are hand-written gfx950 HIP kernels (``csrc/kernels``), gradient synchronisation is
a native bucketed reducer on RCCL (``csrc/comm``).  See SURVEY.md for the map.
"""
__version__ = "0.1.0"

from . import backend  # noqa: F401
