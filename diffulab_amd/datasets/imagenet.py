"""``diffulab.datasets.imagenet`` -- the module the reference's ImageNet entry scripts import (datasets/imagenet.py).

* ``ImageNetLatentREPA`` (imagenet.py:18-86): re-exported from ``.latents`` (class-conditional latents + REPA target features).
* ``ImageNetmultiAR`` (imagenet.py:89-175): the text-to-image dataset of ``examples/train_repa_txt_to_img.py`` -- precomputed VAE
  latents of several aspect ratios with a caption each; the samples are grouped into BUCKETS keyed by the image's (height, width)
  (imagenet.py:109-123, cached as a pickle under ``~/.cache/diffulab``), and a batch only ever holds samples of one bucket.
* ``collate_fn`` (imagenet.py:177-194) and ``MultiARBatchSampler`` (imagenet.py:197-236): the loader pieces that script passes to
  ``torch.utils.data.DataLoader``.  The sampler is integer / index work on Python's ``random`` module: with the same seed it yields
  the reference's batch lists bit for bit (``tests/golden/multiar.npz``, generated from the imported reference).

The shards are read by ``.mds`` (mosaicml-streaming is not installed here).  The bucket key of a sample is the size of its ``image``
column when the shards carry one (only the image HEADER is parsed: MDS ``pil`` / ``jpeg`` / ``png`` encodings); shards that hold
only the precomputed columns are bucketed by the latent's own (height, width) -- the same partition of the indices whenever the
latents were computed from bucket-sized images, which is what the reference's offline pipeline does (imagenet.py:136: "everything
needs to be done offline").  The ``x0 = ToTensor()(image)`` branch for shards WITHOUT ``dst_features`` (imagenet.py:167-170) feeds
a REPA image encoder, which is out of scope: such shards are refused with that reason.
"""

from __future__ import annotations

import logging
import math
import pickle
import random
from pathlib import Path
from typing import Any, Iterator

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler

from .base import BatchData
from .latents import ImageNetLatentREPA

__all__ = ["ImageNetLatentREPA", "ImageNetmultiAR", "MultiARBatchSampler", "collate_fn"]


class ImageNetmultiAR(Dataset):
    def __init__(self, data_path: str, local: bool = True, batch_size: int = 64, split: str = "train",
                 cache_dir: str | Path | None = None) -> None:
        """same arguments as the reference (imagenet.py:90-96); ``cache_dir`` (extra, default ``~/.cache/diffulab`` as in the
        reference) is where ``buckets_cache_imagenet_<split>.pickle`` lives"""
        super().__init__()
        if not local:
            raise NotImplementedError("remote (streaming) shards are not supported: copy the split locally")
        from .mds import MDSDataset

        self.latent_scale: float | None = None
        self.latent_bias: float = 0.0
        self.data_path = Path(data_path)
        self.batch_size = batch_size  # (a streaming hint in the reference; unused by a local reader)
        root = self.data_path / split
        self.dataset = MDSDataset(self.data_path, split if (root / "index.json").exists() else None,
                                  columns=("vision_latents", "caption", "dst_features"))
        names = set(self.dataset.column_names)
        missing = {"vision_latents", "caption"} - names
        if missing:  # (the reference asserts this per item, imagenet.py:147-150)
            raise ValueError(f"{self.dataset.root}: the MDS shards lack the column(s) {sorted(missing)}: precompute the latents / add captions first")
        if "dst_features" not in names:
            raise NotImplementedError(f"{self.dataset.root}: no 'dst_features' column -- the reference then hands the raw 'image' to a "
                                      "REPA encoder (imagenet.py:167-170), which is out of scope here: precompute the features")
        cache = (Path(cache_dir) if cache_dir is not None else Path.home() / ".cache" / "diffulab") / f"buckets_cache_imagenet_{split}.pickle"
        if not cache.exists():
            logging.info("No buckets cache found, constructing buckets...")
            self.buckets: dict[tuple[int, int], list[int]] = {}
            for b in range(len(self.dataset)):  # (dataset order, like the reference's enumerate over the stream)
                self.buckets.setdefault(self._bucket_key(b), []).append(b)
            cache.parent.mkdir(parents=True, exist_ok=True)
            with open(cache, "wb") as f:
                pickle.dump(self.buckets, f)
        else:
            logging.info("Loading buckets from cache...")
            with open(cache, "rb") as f:
                self.buckets = pickle.load(f)

    def _bucket_key(self, idx: int) -> tuple[int, int]:
        """(height, width) of the sample's image (imagenet.py:113-116), or of its latent when the shards carry no image column"""
        if "image" in self.dataset.column_names:
            return self.dataset.image_size(idx, "image")
        lat = self.dataset.get(idx, ("vision_latents",))["vision_latents"]
        return (int(lat.shape[-2]), int(lat.shape[-1]))

    def __len__(self) -> int:
        return sum(len(v) for v in self.buckets.values())

    def set_latent_scale(self, scale: float) -> None:
        self.latent_scale = scale

    def set_latent_bias(self, bias: float) -> None:
        self.latent_bias = bias

    def __getitem__(self, idx: int) -> BatchData:
        assert self.latent_scale is not None, "Latent scale must be set before getting items"
        smp = self.dataset[idx]
        latent = torch.tensor(np.asarray(smp["vision_latents"]), dtype=torch.float32)
        return {  # imagenet.py:152-175
            "model_inputs": {"x": ((latent - self.latent_bias) * self.latent_scale).squeeze(), "initial_context": smp["caption"]},
            "extra": {"dst_features": torch.tensor(np.asarray(smp["dst_features"]), dtype=torch.float32)},
        }


def collate_fn(batch: list[BatchData]) -> BatchData:
    """imagenet.py:177-194: tensors of ``model_inputs`` are stacked, ``initial_context`` stays a list of strings (a missing one
    becomes ""), the ``extra`` tensors are stacked over the samples that HAVE the key"""
    model_inputs: dict[str, Any] = {}
    for key in batch[0]["model_inputs"].keys():
        if key == "initial_context":
            model_inputs[key] = [s["model_inputs"].get(key, "") for s in batch]
        else:
            model_inputs[key] = torch.stack([s["model_inputs"][key] for s in batch], dim=0)
    extra: dict[str, Any] = {}
    for key in set().union(*(s.get("extra", {}).keys() for s in batch)):
        extra[key] = torch.stack([s["extra"][key] for s in batch if key in s.get("extra", {})], dim=0)
    return {"model_inputs": model_inputs, "extra": extra}  # type: ignore[return-value]


class MultiARBatchSampler(Sampler):
    """imagenet.py:197-236.  Every batch comes from ONE aspect-ratio bucket: per bucket (dict order) the indices are shuffled with
    ``random.shuffle`` and cut into batches (a short last one is dropped with ``drop_last``), then the list of all batches is
    shuffled once more -- two kinds of draws on the global ``random`` state, in that order, so ``random.seed(s)`` reproduces the
    reference's epoch exactly."""

    def __init__(self, dataset: ImageNetmultiAR, batch_size: int, shuffle: bool = True, drop_last: bool = False) -> None:
        if not hasattr(dataset, "buckets"):
            raise ValueError("Dataset must have 'buckets' attribute for MultiARBatchSampler")
        self.shuffle = shuffle
        self.buckets = dataset.buckets
        self.batch_size = batch_size
        self.drop_last = drop_last

    def __iter__(self) -> Iterator[list[int]]:
        all_batches: list[list[int]] = []
        for idxs in self.buckets.values():
            idxs = idxs.copy()
            if self.shuffle:
                random.shuffle(idxs)
            for i in range(0, len(idxs), self.batch_size):
                batch = idxs[i : i + self.batch_size]
                if len(batch) < self.batch_size and self.drop_last:
                    continue
                all_batches.append(batch)
        if self.shuffle:
            random.shuffle(all_batches)
        yield from all_batches

    def __len__(self) -> int:
        if self.drop_last:
            return sum(len(v) // self.batch_size for v in self.buckets.values())
        return sum(math.ceil(len(v) / self.batch_size) for v in self.buckets.values())
