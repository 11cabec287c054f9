"""diffulab_amd -- MI355X-native (gfx950) build of DiffuLab's denoising hot path.

Same plugin API as the reference (`Diffuser` / `Denoiser` / `Sampler`), hand-written HIP underneath
(`libdiffulab_hip.so`, C ABI in include/diffulab_hip.h).  See DESIGN.md.
"""

from . import ops  # noqa: F401
from .diffuse import Diffuser, Flow, GaussianDiffusion  # noqa: F401
from .networks.denoisers import DDT, Denoiser, MMDiT, SprintDiT, UNetModel  # noqa: F401

__all__ = ["Diffuser", "Flow", "GaussianDiffusion", "DDT", "Denoiser", "MMDiT", "SprintDiT", "UNetModel", "ops"]
