"""Batch carrier of the training loop (mirrors datasets/base.py:13-15 of the reference)."""

from __future__ import annotations

from typing import TypedDict

from torch import Tensor

try:  # python >= 3.11
    from typing import NotRequired, Required
except ImportError:  # python 3.10 (this image)
    from typing_extensions import NotRequired, Required

from ..networks.denoisers.common import ModelInput


class BatchData(TypedDict, total=False):
    model_inputs: Required[ModelInput]
    extra: NotRequired[dict[str, Tensor | list[str] | None]]
