"""Batch carrier of the training loop (mirrors datasets/base.py:13-15 of the reference)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import TypedDict

import torch
from torch import Tensor
from torch.utils.data import Dataset

try:  # python >= 3.11
    from typing import NotRequired, Required
except ImportError:  # python 3.10 (this image)
    from typing_extensions import NotRequired, Required

from ..networks.denoisers.common import ModelInput


class BatchData(TypedDict, total=False):
    model_inputs: Required[ModelInput]
    extra: NotRequired[dict[str, Tensor | list[str] | None]]


class BaseDataset(Dataset, ABC):
    """image / label dataset base (mirrors datasets/base.py:23-75 of the reference): subclasses fill ``images`` / ``labels`` in
    ``load_data`` and normalise one sample in ``preprocess_image``; an item is ``{"model_inputs": {"x": image, "y": label}}``"""

    def __init__(self) -> None:
        super().__init__()
        self.images = None
        self.labels = None

    @abstractmethod
    def load_data(self): ...

    @abstractmethod
    def preprocess_image(self, image): ...

    def __len__(self) -> int:
        if self.images is None:
            raise ValueError("Dataset has not been initialized properly. Images are None.")
        return len(self.images)

    def __getitem__(self, idx: int) -> BatchData:
        if self.images is None or self.labels is None:
            raise ValueError("Dataset has not been initialized properly. Images or labels are None.")
        return {"model_inputs": {"x": torch.tensor(self.preprocess_image(self.images[idx])),
                                 "y": torch.tensor(self.labels[idx], dtype=torch.long)}}
