"""diffulab_amd -- MI355X-native (gfx950) build of DiffuLab's denoising hot path.

Same plugin API as the reference (`Diffuser` / `Denoiser` / `Sampler`), hand-written HIP underneath
(`libdiffulab_hip.so`, C ABI in include/diffulab_hip.h).  See DESIGN.md.
"""

from . import ops  # noqa: F401
from .datasets import BaseDataset, CIFAR10Dataset, ImageNetLatentREPA, MNISTDataset, SyntheticDataset  # noqa: F401
from .diffuse import Diffuser, Flow, GaussianDiffusion  # noqa: F401
from .networks import PerceiverResampler, PrecomputedEmbedder  # noqa: F401
from .networks.denoisers import DDT, Denoiser, MMDiT, SprintDiT, UNetModel  # noqa: F401
from .training import BaseTrainer, LossFunction, RepaLoss, Trainer  # noqa: F401

# (the names of the reference's top-level package that lie on the hot path; VAEs / text encoders / DINO / GRPO are out of scope)
__all__ = ["BaseDataset", "CIFAR10Dataset", "ImageNetLatentREPA", "MNISTDataset", "SyntheticDataset", "Diffuser", "Flow",
           "GaussianDiffusion", "DDT", "Denoiser", "MMDiT", "SprintDiT", "UNetModel", "PerceiverResampler", "PrecomputedEmbedder",
           "BaseTrainer", "Trainer", "LossFunction", "RepaLoss", "ops"]
