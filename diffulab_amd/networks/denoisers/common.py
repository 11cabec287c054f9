"""Denoiser plugin interface (mirrors networks/denoisers/common.py:8-46 of the reference)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, TypedDict

import torch.nn as nn
from torch import Tensor

try:  # python >= 3.11
    from typing import NotRequired, Required
except ImportError:  # python 3.10 (this image)
    from typing_extensions import NotRequired, Required


class ModelInput(TypedDict, total=False):
    x: Required[Tensor]
    p: NotRequired[float]            # label-drop probability (classifier-free guidance)
    y: NotRequired[Tensor]           # class labels
    initial_context: NotRequired[Any]
    x_context: NotRequired[Tensor]   # concatenated to x along channels


class ModelOutput(TypedDict, total=False):
    x: Required[Tensor]
    features: NotRequired[list[Tensor]]
    repa_features: NotRequired[list[Tensor]]


class Denoiser(nn.Module, ABC):
    classifier_free: bool

    def __init__(self) -> None:
        super().__init__()

    @abstractmethod
    def forward(self, x: Tensor, timesteps: Tensor, *args: Any, **kwargs: Any) -> ModelOutput: ...
