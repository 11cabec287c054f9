"""Context-embedder plugin interface (mirrors networks/embedders/common.py:1-64 of the reference)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, TypedDict

import torch.nn as nn
from torch import Tensor

try:  # python >= 3.11
    from typing import NotRequired, Required
except ImportError:  # python 3.10 (this image)
    from typing_extensions import NotRequired, Required


class ContextEmbedderOutput(TypedDict):
    embeddings: Required[Tensor]
    pooled_embeddings: NotRequired[Tensor]
    attn_mask: NotRequired[Tensor]


class ContextEmbedder(nn.Module, ABC):
    _n_output: int
    _output_size: tuple[int, ...]

    def __init__(self) -> None:
        super().__init__()

    @property
    def n_output(self) -> int:
        """number of embeddings the embedder returns"""
        return self._n_output

    @property
    def output_size(self) -> tuple[int, ...]:
        """width of each returned embedding"""
        return self._output_size

    @abstractmethod
    def drop_conditions(self, context: Any, p: float) -> Any: ...

    @abstractmethod
    def forward(self, context: Any, p: float = 0) -> ContextEmbedderOutput: ...
