"""Reader of uncompressed MosaicML MDS shards -- the on-disk format behind ``streaming.StreamingDataset(local=...)``, which is what
the reference's ``ImageNetLatentREPA`` reads (reference datasets/imagenet.py:18-41; columns ``vision_latents`` / ``label`` /
``dst_features``, imagenet.py:62-86; written with ``MDSWriter(columns={..., "vision_latents": "ndarray:float32"})`` in
networks/vision_towers/common.py:137-151 and ``"dst_features": "ndarray:<float32|float16>"`` in networks/repa/common.py:96-111).

mosaicml-streaming (pinned 0.13.0 in the reference's uv.lock) is NOT installed in this image and there is no network, so the
format is restated here from its published layout; PARITY UNPINNED against the library itself -- ``tests/test_datasets.py`` pins
the byte layout below with hand-assembled shards, and the oracle-side writer (``oracle/synth.py::write_mds``) is an independent
restatement of the same layout:

  <dir>/index.json     {"version": 2, "shards": [{"format": "mds", "column_names": [...], "column_encodings": [...],
                        "column_sizes": [int | null, ...], "compression": null, "samples": S,
                        "raw_data": {"basename": "shard.00000.mds", "bytes": N, "hashes": {}}, "zip_data": null, ...}, ...]}
  shard file           uint32 S | uint32 offset[S + 1] (byte positions of the samples from the start of the file) |
                       the shard's JSON header (skipped: offset[0] points behind it) | sample 0 | sample 1 | ...
  sample               uint32 size of every VARIABLE-size column (column order) | the columns' encoded values (column order)
  encodings            "int": int64 (8 bytes); numpy scalars "uint8" ... "int64", "float16/32/64": that many bytes;
                       "str": utf-8; "bytes": raw; "json": utf-8 JSON;
                       "ndarray[:dtype[:d0,d1,...]]": [uint8 dtype code unless in the header] [unless the shape is in the header:
                       uint8 code c of the shape's integer type (uint8/16/32/64 = 0..3), uint8 ndim, ndim x (dim - 1) in that
                       type] raw little-endian element data.  dtype codes: uint8, uint16, uint32, uint64, int8, int16, int32,
                       int64, float16, float32, float64 = 0..10.
Everything is little-endian.  Compressed shards ("compression" / "zip_data" set) are refused; image codecs ("pil", "jpeg", "png") are never DECODED (asking for
such a column raises) -- only ``image_size`` reads their header, for the aspect-ratio buckets of ``ImageNetmultiAR``: recompress / re-encode with the streaming package, or export the three columns as ``.npy``
(``ImageNetLatentREPA.write_split``).  Columns that are not asked for are skipped without decoding, so a shard that also carries
an ``image`` column is readable."""

from __future__ import annotations

import json
from bisect import bisect_right
from pathlib import Path
from typing import Any

import numpy as np

_ND_DTYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64, np.float16, np.float32, np.float64]
_SCALARS = {np.dtype(t).name: np.dtype(t) for t in _ND_DTYPES}


def _decode(encoding: str, data: bytes) -> Any:
    kind, _, rest = encoding.partition(":")
    if kind == "int":
        return int(np.frombuffer(data, "<i8")[0])
    if kind in _SCALARS:
        return np.frombuffer(data, _SCALARS[kind].newbyteorder("<"))[0]
    if kind == "str":
        return data.decode("utf-8")
    if kind == "bytes":
        return bytes(data)
    if kind == "json":
        return json.loads(data.decode("utf-8"))
    if kind == "ndarray":
        dtype_s, _, shape_s = rest.partition(":")
        i = 0
        if dtype_s:
            dtype = np.dtype(dtype_s)
        else:
            dtype, i = np.dtype(_ND_DTYPES[data[0]]), 1
        if shape_s:
            shape = tuple(int(d) for d in shape_s.split(","))
        else:
            sdt = np.dtype(_ND_DTYPES[data[i]])
            if sdt.kind != "u":
                raise ValueError(f"MDS ndarray: shape type code {data[i]} is not an unsigned integer type")
            ndim = data[i + 1]
            i += 2
            shape = tuple(int(d) + 1 for d in np.frombuffer(data[i : i + ndim * sdt.itemsize], sdt.newbyteorder("<")))
            i += ndim * sdt.itemsize
        return np.frombuffer(data[i:], dtype.newbyteorder("<")).reshape(shape)
    raise NotImplementedError(f"MDS column encoding {encoding!r} is not supported by this reader (supported: int, numpy scalars, str, "
                              "bytes, json, ndarray[:dtype[:shape]])")


class _Shard:
    def __init__(self, root: Path, info: dict) -> None:
        if info.get("format") != "mds":
            raise NotImplementedError(f"shard format {info.get('format')!r}: only 'mds' is supported")
        if info.get("compression") or info.get("zip_data"):
            raise NotImplementedError(f"{root}: compressed MDS shards ({info.get('compression')}) are not supported: decompress them "
                                      "with the streaming package or export the columns as .npy")
        self.names: list[str] = list(info["column_names"])
        self.encodings: list[str] = list(info["column_encodings"])
        self.sizes: list[int | None] = list(info["column_sizes"])
        self.samples = int(info["samples"])
        self.path = root / info["raw_data"]["basename"]
        self._mm: np.memmap | None = None
        self._off: np.ndarray | None = None

    def _open(self) -> None:
        self._mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        n = int(np.frombuffer(self._mm[:4], "<u4")[0])
        if n != self.samples:
            raise ValueError(f"{self.path}: the shard holds {n} samples, index.json says {self.samples}")
        self._off = np.frombuffer(self._mm[4 : 4 + 4 * (n + 1)], "<u4")

    def sample_raw(self, i: int, column: str) -> bytes:
        """the undecoded bytes of one column of sample ``i``"""
        if self._mm is None:
            self._open()
        data = bytes(self._mm[int(self._off[i]) : int(self._off[i + 1])])
        pos, sizes = 0, []
        for size in self.sizes:
            if size is None:
                sizes.append(int(np.frombuffer(data[pos : pos + 4], "<u4")[0]))
                pos += 4
            else:
                sizes.append(int(size))
        for name, size in zip(self.names, sizes):
            if name == column:
                return data[pos : pos + size]
            pos += size
        raise KeyError(column)

    def sample(self, i: int, columns: tuple[str, ...] | None) -> dict[str, Any]:
        if self._mm is None:
            self._open()
        data = bytes(self._mm[int(self._off[i]) : int(self._off[i + 1])])
        pos, sizes = 0, []
        for size in self.sizes:
            if size is None:
                sizes.append(int(np.frombuffer(data[pos : pos + 4], "<u4")[0]))
                pos += 4
            else:
                sizes.append(int(size))
        out: dict[str, Any] = {}
        for name, enc, size in zip(self.names, self.encodings, sizes):
            if columns is None or name in columns:
                out[name] = _decode(enc, data[pos : pos + size])
            pos += size
        if pos != len(data):
            raise ValueError(f"{self.path}: sample {i} has {len(data)} bytes, its columns account for {pos}")
        return out


class MDSDataset:
    """``len()`` / ``[idx]`` over the shards of ``<local>/<split>`` (or ``<local>`` without a split), like an un-shuffled
    ``StreamingDataset(local=..., split=...)``.  ``columns``: decode only these (None: all)."""

    def __init__(self, local: str | Path, split: str | None = None, columns: tuple[str, ...] | None = None) -> None:
        self.root = Path(local) / split if split else Path(local)
        index = self.root / "index.json"
        if not index.exists():
            raise FileNotFoundError(f"{index}: not an MDS directory")
        meta = json.loads(index.read_text())
        if meta.get("version") != 2:
            raise NotImplementedError(f"{index}: MDS index version {meta.get('version')!r} (this reader knows version 2)")
        self.shards = [_Shard(self.root, s) for s in meta["shards"]]
        self._starts = np.cumsum([0] + [s.samples for s in self.shards]).tolist()
        self.columns = columns
        self.column_names = self.shards[0].names if self.shards else []

    def __len__(self) -> int:
        return self._starts[-1]

    def __getitem__(self, idx: int) -> dict[str, Any]:
        n = len(self)
        if idx < 0:
            idx += n
        if not 0 <= idx < n:
            raise IndexError(idx)
        s = bisect_right(self._starts, idx) - 1
        return self.shards[s].sample(idx - self._starts[s], self.columns)

    def get(self, idx: int, columns: tuple[str, ...]) -> dict[str, Any]:
        """sample ``idx`` decoding only ``columns`` (whatever the dataset's own column filter is)"""
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        s = bisect_right(self._starts, idx) - 1
        return self.shards[s].sample(idx - self._starts[s], columns)

    def image_size(self, idx: int, column: str = "image") -> tuple[int, int]:
        """(height, width) of an image column of sample ``idx`` from the image's HEADER only (``pil`` / ``jpeg`` / ``png``
        encodings; the pixels are never decoded) -- what the multi-aspect-ratio dataset buckets by (reference imagenet.py:113-116:
        ``w, h = sample["image"].size``)"""
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        s = bisect_right(self._starts, idx) - 1
        sh = self.shards[s]
        enc = sh.encodings[sh.names.index(column)]
        raw = sh.sample_raw(idx - self._starts[s], column)
        return image_size_of(enc, raw)


def image_size_of(encoding: str, data: bytes) -> tuple[int, int]:
    """(height, width) from the first bytes of an MDS image column.  ``pil``: uint32 width | uint32 height | uint32 len(mode) | mode |
    raw pixels (the streaming package's PIL encoding); ``png``: the IHDR chunk; ``jpeg``: the first start-of-frame marker."""
    kind = encoding.partition(":")[0]
    if kind == "pil":
        w, h = (int(v) for v in np.frombuffer(data[:8], "<u4"))
        return h, w
    if kind == "png" or data[:8] == b"\x89PNG\r\n\x1a\n":
        if data[:8] != b"\x89PNG\r\n\x1a\n" or data[12:16] != b"IHDR":
            raise ValueError("MDS png column: not a PNG stream")
        return int.from_bytes(data[20:24], "big"), int.from_bytes(data[16:20], "big")
    if kind == "jpeg" or data[:2] == b"\xff\xd8":
        if data[:2] != b"\xff\xd8":
            raise ValueError("MDS jpeg column: not a JPEG stream")
        i = 2
        while i + 9 < len(data):
            if data[i] != 0xFF:
                raise ValueError("MDS jpeg column: lost marker synchronisation")
            m = data[i + 1]
            if m == 0xFF:  # fill byte
                i += 1
                continue
            if 0xD0 <= m <= 0xD9 or m == 0x01:  # standalone markers
                i += 2
                continue
            seg = int.from_bytes(data[i + 2 : i + 4], "big")
            if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):  # SOFn: precision | height | width
                return int.from_bytes(data[i + 5 : i + 7], "big"), int.from_bytes(data[i + 7 : i + 9], "big")
            i += 2 + seg
        raise ValueError("MDS jpeg column: no start-of-frame marker")
    raise NotImplementedError(f"image size of MDS column encoding {encoding!r} (supported: pil, png, jpeg)")
