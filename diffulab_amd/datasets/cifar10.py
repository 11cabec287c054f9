"""CIFAR-10 from the python-pickle batches (reference datasets/cifar10.py:10-85: same constructor / item format / arithmetic).

A batch file is a latin1 pickle with ``data`` uint8 [N, 3072] (planar R, G, B of 32x32) and ``labels``; images are kept as uint8
HWC and a sample is normalised to [-1, 1] and transposed to CHW on access."""

from __future__ import annotations

import pickle
from pathlib import Path

import numpy as np

from .base import BaseDataset

_ALL = ("data_batch_1", "data_batch_2", "data_batch_3", "data_batch_4", "data_batch_5")


class CIFAR10Dataset(BaseDataset):
    def __init__(self, data_path: str, batches_to_load: list[str] | tuple[str, ...] = _ALL) -> None:
        super().__init__()
        self.data_path = Path(data_path)
        self.batches_to_load = list(batches_to_load)
        self.images, self.labels = self.load_data()

    def load_data(self) -> tuple[np.ndarray, np.ndarray]:
        parts = [self._load_cifar10_batch(self.data_path / b) for b in self.batches_to_load]
        return np.concatenate([p[0] for p in parts], axis=0), np.concatenate([p[1] for p in parts], axis=0)

    @staticmethod
    def _load_cifar10_batch(file: Path) -> tuple[np.ndarray, np.ndarray]:
        with open(file, "rb") as f:
            batch = pickle.load(f, encoding="latin1")
        planes = np.asarray(batch["data"], dtype=np.uint8).reshape(-1, 3, 32, 32)  # planar RGB
        return np.ascontiguousarray(planes.transpose(0, 2, 3, 1)), np.asarray(batch["labels"], dtype=np.int64)

    def preprocess_image(self, image: np.ndarray) -> np.ndarray:
        return ((image.astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(2, 0, 1)
