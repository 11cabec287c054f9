"""Sampler plugin interface (reference diffuse/samplers/common.py:7-32)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, TypedDict

from torch import Tensor

try:
    from typing import NotRequired, Required
except ImportError:
    from typing_extensions import NotRequired, Required


class StepResult(TypedDict):
    x_prev: Required[Tensor]
    estimated_x0: Required[Tensor]
    x_prev_mean: NotRequired[Tensor]
    x_prev_std: NotRequired[Tensor]
    logprob: NotRequired[Tensor]


class Sampler(ABC):
    name: str

    def __init__(self) -> None:
        pass

    @abstractmethod
    def set_steps(self, *args: Any, **kwargs: Any) -> None: ...

    @abstractmethod
    def step(self, *args: Any, **kwargs: Any) -> StepResult: ...
