"""ctypes mirror of ``dl_dit_block_t`` (include/diffulab_hip.h): the argument block of the native DiT block driver
``dl_dit_block_fwd`` / ``dl_dit_block_bwd``.  The slot indices are parsed from the header's enum, so the two sides cannot drift."""

from __future__ import annotations

import ctypes
import re
from functools import lru_cache

from ._lib import HEADER_PATH


@lru_cache(maxsize=1)
def slots() -> dict[str, int]:
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER_PATH).read(), flags=re.S)
    body = text[text.index("DL_BLK_X_IN") : text.index("DL_BLK_NPTR")]
    names = re.findall(r"DL_BLK_([A-Z0-9_]+)", "DL_BLK_" + body.split("DL_BLK_", 1)[1])
    out: dict[str, int] = {}
    for n in names:
        if n not in out:
            out[n] = len(out)
    return out


class DitBlock(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p * 96)] + [(n, ctypes.c_int64) for n in
                ("B", "N", "D", "H", "F", "ld_mod", "ld_dmod", "ldw_d", "ldw_f", "ldwt_d", "ldwt_f2", "ldwt_3d", "rot")] + [
                    ("eps", ctypes.c_float), ("next_eps", ctypes.c_float), ("row_gemms", ctypes.c_int32), ("tn_slab_floats", ctypes.c_int64), ("max_workgroups", ctypes.c_int32)]

    def set(self, **ptrs) -> "DitBlock":
        """slot name (lower case) -> tensor / int address / None"""
        idx = slots()
        for k, v in ptrs.items():
            addr = None if v is None else (v if isinstance(v, int) else v.data_ptr())
            self.p[idx[k.upper()]] = addr
        return self
