"""Precomputed-latent dataset of the ImageNet configs (reference datasets/imagenet.py:18-86, ``ImageNetLatentREPA``).

The reference streams MosaicML MDS shards with the columns ``vision_latents``, ``label`` and ``dst_features`` (or ``image``).
Two on-disk layouts are read:

  * the reference's own: uncompressed MDS shards under ``<data_path>/<split>/`` (``index.json`` + ``shard.*.mds``), through
    ``diffulab_amd/datasets/mds.py`` (the mosaicml-streaming package is absent here; its format is restated there);
  * the same three columns as plain ``.npy`` arrays, memory-mapped (a dump of the shards; what round 1-2 read):

    <data_path>/<split>/vision_latents.npy   f32/f16 [N, C, H, W]   VAE latents (unscaled)
    <data_path>/<split>/label.npy            int     [N]
    <data_path>/<split>/dst_features.npy     f32/f16 [N, T, F]      optional: REPA target features (DINOv2 tokens)

(``ImageNetLatentREPA.write_split`` produces that layout, e.g. from an MDS dump.)  Constructor, ``set_latent_scale`` and the item
format are the reference's: ``{"model_inputs": {"x": latent * latent_scale, "y": label}, "extra": {"dst_features": ...}}``; asking for
an item before ``set_latent_scale`` raises, like the reference's assert."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from .base import BatchData


class ImageNetLatentREPA(Dataset):
    def __init__(self, data_path: str, local: bool = True, batch_size: int = 64, split: str = "train") -> None:
        super().__init__()
        if not local:
            raise NotImplementedError("remote (streaming) shards are not supported: copy the split locally")
        self.data_path = Path(data_path)
        root = self.data_path / split
        self.batch_size = batch_size  # (a streaming hint in the reference; unused by a local reader)
        self.latent_scale: float | None = None
        self.mds = None
        if (root / "index.json").exists() or (not (root / "vision_latents.npy").exists() and (self.data_path / "index.json").exists()):
            from .mds import MDSDataset

            self.mds = MDSDataset(self.data_path, split if (root / "index.json").exists() else None,
                                  columns=("vision_latents", "label", "dst_features"))
            missing = {"vision_latents", "label"} - set(self.mds.column_names)
            if missing:  # (the reference asserts this per item, imagenet.py:64-65)
                raise ValueError(f"{self.mds.root}: the MDS shards lack the column(s) {sorted(missing)}: precompute the latents first")
            if "dst_features" not in self.mds.column_names:
                raise NotImplementedError(f"{self.mds.root}: no 'dst_features' column -- the reference then hands the raw 'image' to a "
                                          "REPA encoder (imagenet.py:76-79), which is out of scope here: precompute the features")
            return
        self.latents = np.load(root / "vision_latents.npy", mmap_mode="r")
        self.labels = np.load(root / "label.npy", mmap_mode="r")
        feats = root / "dst_features.npy"
        self.dst_features = np.load(feats, mmap_mode="r") if feats.exists() else None
        if len(self.labels) != len(self.latents) or (self.dst_features is not None and len(self.dst_features) != len(self.latents)):
            raise ValueError(f"{root}: the columns have different lengths")

    def set_latent_scale(self, scale: float) -> None:
        self.latent_scale = scale

    def __len__(self) -> int:
        return len(self.mds) if self.mds is not None else len(self.latents)

    def __getitem__(self, idx: int) -> BatchData:
        assert self.latent_scale is not None, "Latent scale must be set before getting items"
        if self.mds is not None:  # imagenet.py:62-86: tensors of the three columns, latent scaled
            smp = self.mds[idx]
            latent = torch.tensor(np.asarray(smp["vision_latents"]), dtype=torch.float32)
            return {"model_inputs": {"x": latent * self.latent_scale, "y": torch.tensor(smp["label"], dtype=torch.long)},
                    "extra": {"dst_features": torch.tensor(np.asarray(smp["dst_features"]), dtype=torch.float32)}}
        latent = torch.from_numpy(np.array(self.latents[idx], dtype=np.float32))
        item: BatchData = {"model_inputs": {"x": latent * self.latent_scale, "y": torch.tensor(int(self.labels[idx]), dtype=torch.long)},
                           "extra": {}}
        if self.dst_features is not None:
            item["extra"]["dst_features"] = torch.from_numpy(np.array(self.dst_features[idx], dtype=np.float32))
        return item

    @staticmethod
    def write_split(data_path: str | Path, split: str, vision_latents: np.ndarray, label: np.ndarray,
                    dst_features: np.ndarray | None = None) -> None:
        root = Path(data_path) / split
        root.mkdir(parents=True, exist_ok=True)
        np.save(root / "vision_latents.npy", np.asarray(vision_latents))
        np.save(root / "label.npy", np.asarray(label))
        if dst_features is not None:
            np.save(root / "dst_features.npy", np.asarray(dst_features))
