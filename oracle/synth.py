"""Deterministic synthetic tensors shared by the fixture generator and the tests.  TEST INFRASTRUCTURE ONLY.

Fixtures store only the reference's OUTPUTS; inputs and weights are regenerated on
either box from numpy's PCG64 stream (same numpy in the build container and on the
GPU box), keyed by a string so that adding a tensor never shifts another one.
"""

from __future__ import annotations

import zlib

import numpy as np
import torch
from torch import Tensor


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def normal(key: str, shape, seed: int = 0, std: float = 1.0) -> Tensor:
    a = _rng(key, seed).standard_normal(tuple(shape)).astype(np.float32) * np.float32(std)
    return torch.from_numpy(a)


def uniform(key: str, shape, seed: int = 0, lo: float = 0.0, hi: float = 1.0) -> Tensor:
    a = _rng(key, seed).random(tuple(shape), dtype=np.float64) * (hi - lo) + lo
    return torch.from_numpy(a.astype(np.float32))


def integers(key: str, shape, hi: int, seed: int = 0) -> Tensor:
    return torch.from_numpy(_rng(key, seed).integers(0, hi, tuple(shape), dtype=np.int64))


def dit_params(shapes: dict[str, tuple[int, ...]], seed: int = 0, mod_std: float = 0.02) -> dict[str, Tensor]:
    """Non-degenerate DiT weights (the reference's zero-init of the adaLN layers would turn every
    block into the identity and hide bugs): fan-in scaled normals for matrices, O(1) norm
    weights around 1, small non-zero biases / modulation rows.
    """
    out: dict[str, Tensor] = {}
    for name, shp in shapes.items():
        if name.endswith(".scale") or name.endswith("norm_1.weight") or name.endswith("norm_2.weight"):
            out[name] = 1.0 + normal(name, shp, seed, 0.1)
        elif name.endswith(".bias"):
            out[name] = normal(name, shp, seed, 0.05)
        elif "modulation" in name or "adaLN" in name:
            out[name] = normal(name, shp, seed, mod_std * 4)
        elif name.endswith("embedding.weight"):
            out[name] = normal(name, shp, seed, 0.5)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[name] = normal(name, shp, seed, fan_in**-0.5)
    return out


def generic_params(shapes: dict[str, tuple[int, ...]], seed: int = 0) -> dict[str, Tensor]:
    """Non-degenerate weights for any module tree (used for the UNet): 1-D ``weight`` -> norm gain around 1, ``bias``
    -> small, ``embedding.weight`` -> N(0, 0.5), everything else -> N(0, 1/fan_in) (the reference's zero-init convs
    would hide bugs behind exact zeros)."""
    out: dict[str, Tensor] = {}
    for name, shp in shapes.items():
        if name.endswith(".bias"):
            out[name] = normal(name, shp, seed, 0.05)
        elif name.endswith("embedding.weight"):
            out[name] = normal(name, shp, seed, 0.5)
        elif len(shp) == 1:
            out[name] = 1.0 + normal(name, shp, seed, 0.1)
        else:
            out[name] = normal(name, shp, seed, int(np.prod(shp[1:])) ** -0.5)
    return out


def write_fake_mnist(root: str, n_train: int = 12, n_test: int = 5, seed: int = 11) -> None:
    """seeded digits in the REAL idx on-disk format (big-endian magic / counts, uint8 payload): inputs of the dataset-reader tests"""
    import os

    rng = np.random.default_rng(seed)
    for stem, n in (("train", n_train), ("t10k", n_test)):
        img = rng.integers(0, 256, size=(n, 28, 28), dtype=np.uint8)
        lab = rng.integers(0, 10, size=(n,), dtype=np.uint8)
        with open(os.path.join(root, f"{stem}-images-idx3-ubyte"), "wb") as f:
            f.write(np.array([2051, n, 28, 28], dtype=">u4").tobytes() + img.tobytes())
        with open(os.path.join(root, f"{stem}-labels-idx1-ubyte"), "wb") as f:
            f.write(np.array([2049, n], dtype=">u4").tobytes() + lab.tobytes())


def write_fake_cifar10(root: str, batches: dict[str, int], seed: int = 12) -> None:
    """seeded images in the REAL python-pickle batch format of CIFAR-10 (dict with "data" uint8 [N, 3072] and "labels")"""
    import os
    import pickle

    rng = np.random.default_rng(seed)
    for name, n in batches.items():
        d = {"data": rng.integers(0, 256, size=(n, 3072), dtype=np.uint8), "labels": [int(v) for v in rng.integers(0, 10, size=n)],
             "batch_label": name, "filenames": [f"{i}.png" for i in range(n)]}
        with open(os.path.join(root, name), "wb") as f:
            pickle.dump(d, f, protocol=2)


def write_mds(root: str, columns: dict[str, str], samples: list[dict], shard_samples: int = 0, extra_header: dict | None = None) -> None:
    """TEST INFRASTRUCTURE: writes `samples` as uncompressed MosaicML MDS shards (index.json + shard.NNNNN.mds) -- an independent
    restatement of the layout streaming.MDSWriter produces (mosaicml-streaming 0.13.0, the reference's pinned version; the package
    is absent here), used to test diffulab_amd/datasets/mds.py.  columns: name -> encoding ("int", "str", "bytes",
    "ndarray[:dtype[:shape]]", numpy scalar names); shard_samples > 0 cuts a new shard every that many samples."""
    import json
    import os

    nd_codes = {np.dtype(t): i for i, t in enumerate([np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32,
                                                      np.int64, np.float16, np.float32, np.float64])}

    def enc(encoding: str, v) -> bytes:
        kind, _, rest = encoding.partition(":")
        if kind == "int":
            return np.int64(v).tobytes()
        if kind == "str":
            return str(v).encode("utf-8")
        if kind in ("bytes", "png", "jpeg", "pil", "pkl"):  # (opaque payloads: the tests hand in the already encoded bytes)
            return bytes(v)
        if kind == "ndarray":
            a = np.ascontiguousarray(v)
            dtype_s, _, shape_s = rest.partition(":")
            parts = []
            if dtype_s:
                assert a.dtype == np.dtype(dtype_s), (a.dtype, dtype_s)
            else:
                parts.append(bytes([nd_codes[a.dtype]]))
            if shape_s:
                assert a.shape == tuple(int(d) for d in shape_s.split(","))
            else:
                big = max(a.shape)
                st = np.uint8 if big <= 1 << 8 else np.uint16 if big <= 1 << 16 else np.uint32 if big <= 1 << 32 else np.uint64
                parts += [bytes([nd_codes[np.dtype(st)]]), bytes([a.ndim]), (np.array(a.shape, np.int64) - 1).astype(st).tobytes()]
            return b"".join(parts) + a.tobytes()
        return np.dtype(kind).type(v).tobytes()  # numpy scalar encodings

    def fixed_size(encoding: str):
        kind, _, rest = encoding.partition(":")
        if kind == "int":
            return 8
        if kind == "ndarray":
            dtype_s, _, shape_s = rest.partition(":")
            return int(np.dtype(dtype_s).itemsize * np.prod([int(d) for d in shape_s.split(",")])) if dtype_s and shape_s else None
        return None if kind in ("str", "bytes", "json", "pkl", "pil", "jpeg", "png") else np.dtype(kind).itemsize

    os.makedirs(root, exist_ok=True)
    names = sorted(columns)  # (MDSWriter sorts the column names)
    encs, sizes = [columns[n] for n in names], [fixed_size(columns[n]) for n in names]
    per = shard_samples if shard_samples > 0 else max(1, len(samples))
    shards = []
    for si, lo in enumerate(range(0, len(samples), per)):
        chunk = samples[lo : lo + per]
        blobs = []
        for smp in chunk:
            data = [enc(e, smp[n]) for n, e in zip(names, encs)]
            head = np.array([len(d) for d, sz in zip(data, sizes) if sz is None], dtype="<u4").tobytes()
            for d, sz in zip(data, sizes):
                assert sz is None or len(d) == sz
            blobs.append(head + b"".join(data))
        info = {"column_encodings": encs, "column_names": names, "column_sizes": sizes, "compression": None, "format": "mds",
                "hashes": [], "size_limit": 1 << 26, "version": 2, **(extra_header or {})}
        config = json.dumps(info, sort_keys=True).encode("utf-8")
        n = np.uint32(len(chunk))
        offsets = np.cumsum([0] + [len(b) for b in blobs]).astype("<u4") + np.uint32(4 + 4 * (len(chunk) + 1) + len(config))
        raw = n.tobytes() + offsets.astype("<u4").tobytes() + config + b"".join(blobs)
        base = f"shard.{si:05d}.mds"
        with open(os.path.join(root, base), "wb") as f:
            f.write(raw)
        shards.append({**info, "raw_data": {"basename": base, "bytes": len(raw), "hashes": {}}, "samples": len(chunk), "zip_data": None})
    with open(os.path.join(root, "index.json"), "w") as f:
        json.dump({"shards": shards, "version": 2}, f, sort_keys=True)


def multiar_buckets() -> dict:
    """the bucket table both sides of the multi-aspect-ratio fixture use: five (height, width) buckets of 23 / 8 / 1 / 16 / 5 indices
    (dict order = first appearance in the dataset, as the reference builds it)"""
    sizes = {(256, 256): 23, (192, 320): 8, (320, 192): 1, (224, 288): 16, (288, 224): 5}
    order = np.random.default_rng(77).permutation(sum(sizes.values())).tolist()
    keys = [k for k, n in sizes.items() for _ in range(n)]
    b: dict = {}
    for idx, pos in enumerate(order):
        b.setdefault(keys[pos], []).append(idx)
    return b
