"""In-memory synthetic dataset with the item format of the reference datasets (datasets/base.py:50-75:
``{"model_inputs": {"x": image, "y": label}}``).  There is no network on the build / GPU boxes, so MNIST / CIFAR / ImageNet
latents are replaced by seeded N(0,1) tensors of the same shape (``torch.Generator().manual_seed(1234)``)."""

from __future__ import annotations

import torch
from torch.utils.data import Dataset

from .base import BatchData


class SyntheticDataset(Dataset):
    def __init__(self, n_samples: int = 1024, shape: tuple[int, ...] | list[int] = (1, 32, 32), n_classes: int | None = 10,
                 seed: int = 1234, dst_features_shape: tuple[int, ...] | list[int] | None = None,
                 context_shape: tuple[int, int] | list[int] | None = None) -> None:
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn(n_samples, *shape, generator=g).clamp_(-3, 3) / 3  # in [-1, 1] like normalised images
        self.labels = torch.randint(0, n_classes, (n_samples,), generator=g) if n_classes is not None else None
        # REPA: precomputed encoder features per sample (datasets/imagenet.py:177-236 ships them as "dst_features")
        self.dst_features = torch.randn(n_samples, *dst_features_shape, generator=g) if dst_features_shape else None
        # text-to-image: precomputed text embeddings [Lc, width] with a ragged validity mask, the input of PrecomputedEmbedder
        # (embedders/precomputed.py:41-43: {"embeddings", "attn_mask"})
        self.context = self.context_mask = None
        if context_shape:
            Lc, width = context_shape
            self.context = torch.randn(n_samples, Lc, width, generator=g)
            self.context_mask = torch.arange(Lc)[None, :] < torch.randint(1, Lc + 1, (n_samples, 1), generator=g)

    def __len__(self) -> int:
        return self.images.shape[0]

    def __getitem__(self, idx: int) -> BatchData:
        inputs = {"x": self.images[idx]}
        if self.labels is not None:
            inputs["y"] = self.labels[idx]
        if self.context is not None:
            inputs["initial_context"] = {"embeddings": self.context[idx], "attn_mask": self.context_mask[idx]}
        item: BatchData = {"model_inputs": inputs}
        if self.dst_features is not None:
            item["extra"] = {"dst_features": self.dst_features[idx]}
        return item
