"""diffulab_amd -- MI355X-native (gfx950) build of DiffuLab's denoising hot path.

Same plugin API as the reference (`Diffuser` / `Denoiser` / `Sampler`), hand-written HIP underneath
(`libdiffulab_hip.so`, C ABI in include/diffulab_hip.h).  See DESIGN.md.
"""

import os as _os

# RCCL between one-process-per-GPU ranks needs dmabuf IPC on hosts whose driver has no legacy IPC (`hipIpcGetMemHandle: invalid
# argument` otherwise).  The HIP runtime reads the variable when it initialises, so it is pinned at import time -- before the first
# GPU call of an ordinary script -- and not in the Trainer (where a script that had already touched the GPU got a silent no-op).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import ops  # noqa: E402,F401
from .datasets import BaseDataset, CIFAR10Dataset, ImageNetLatentREPA, MNISTDataset, SyntheticDataset  # noqa: F401
from .diffuse import Diffuser, Flow, GaussianDiffusion  # noqa: F401
from .networks import PerceiverResampler, PrecomputedEmbedder  # noqa: F401
from .networks.denoisers import DDT, Denoiser, MMDiT, SprintDiT, UNetModel  # noqa: F401
from .training import BaseTrainer, LossFunction, RepaLoss, Trainer  # noqa: F401

# (the names of the reference's top-level package that lie on the hot path; VAEs / text encoders / DINO / GRPO are out of scope)
__all__ = ["BaseDataset", "CIFAR10Dataset", "ImageNetLatentREPA", "MNISTDataset", "SyntheticDataset", "Diffuser", "Flow",
           "GaussianDiffusion", "DDT", "Denoiser", "MMDiT", "SprintDiT", "UNetModel", "PerceiverResampler", "PrecomputedEmbedder",
           "BaseTrainer", "Trainer", "LossFunction", "RepaLoss", "ops"]
