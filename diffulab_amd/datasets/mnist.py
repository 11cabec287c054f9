"""MNIST from the raw idx files (reference datasets/mnist.py:11-86: same constructor, same item format, same arithmetic).

The four ``*-ubyte`` files are read with numpy (big-endian idx header), the 28x28 digits are centred in a zero 32x32 frame and a
sample is normalised to [-1, 1] by ``(x / 255 - 0.5) / 0.5`` -- the padding therefore becomes -1, as in the reference."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from .base import BaseDataset


class MNISTDataset(BaseDataset):
    def __init__(self, data_path: str, train: bool = True) -> None:
        super().__init__()
        self.data_path = Path(data_path)
        self.train = train
        self.images, self.labels = self.load_data()

    def load_data(self) -> tuple[np.ndarray, np.ndarray]:
        stem = "train" if self.train else "t10k"
        return self._load_images(self.data_path / f"{stem}-images-idx3-ubyte"), self._load_labels(self.data_path / f"{stem}-labels-idx1-ubyte")

    @staticmethod
    def _load_images(file: Path) -> np.ndarray:
        raw = np.fromfile(file, dtype=np.uint8)
        magic, n, rows, cols = raw[:16].view(">u4")
        if magic != 2051:
            raise ValueError(f"{file}: not an idx3-ubyte image file (magic {magic})")
        digits = raw[16 : 16 + int(n) * int(rows) * int(cols)].reshape(int(n), 1, int(rows), int(cols))
        framed = np.zeros((int(n), 1, 32, 32), dtype=np.float32)
        r0, c0 = (32 - int(rows)) // 2, (32 - int(cols)) // 2
        framed[:, :, r0 : r0 + int(rows), c0 : c0 + int(cols)] = digits
        return framed

    @staticmethod
    def _load_labels(file: Path) -> np.ndarray:
        raw = np.fromfile(file, dtype=np.uint8)
        magic, n = raw[:8].view(">u4")
        if magic != 2049:
            raise ValueError(f"{file}: not an idx1-ubyte label file (magic {magic})")
        return raw[8 : 8 + int(n)].astype(np.int64)

    def preprocess_image(self, image: np.ndarray) -> np.ndarray:
        return ((image.astype(np.float32) / 255.0) - 0.5) / 0.5
