"""ctypes binding of ``libdiffulab_hip.so`` (the C ABI declared in ``include/diffulab_hip.h``).

The prototypes are parsed from the header itself so the Python side can never drift from the ABI.
There is NO fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is raised
(the product path is the HIP library).
"""

from __future__ import annotations

import ctypes
import os
import re
from functools import lru_cache

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIFFULAB_HIP_LIB") or os.path.join(_HERE, "libdiffulab_hip.so")  # env override: A/B of two builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "diffulab_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "dl_stream_t": ctypes.c_void_p,
}


_VALUE_RETURNING = {"dl_version", "dl_mse_loss_partials"}  # every other int-returning entry point returns a status


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_char_p if re.match(r"^(const\s+)?char\s*\*", decl) and "arch" not in decl else ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _CTYPES[base]


def parse_header(path: str = HEADER_PATH) -> dict[str, tuple[object, list[object], list[str]]]:
    """-> {symbol: (restype, [argtypes], [argnames])} for every ``DL_API`` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos: dict[str, tuple[object, list[object], list[str]]] = {}
    for m in re.finditer(r"DL_API\s+([\w\s\*]+?)\s*\b(dl_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.replace(" ", "") == "constchar*":
            restype = ctypes.c_char_p
        else:
            restype = _CTYPES[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argnames.append(re.split(r"[\s\*]+", a)[-1])
                argtypes.append(_ctype_of(a))
        protos[name] = (restype, argtypes, argnames)
    return protos


class _Lib:
    def __init__(self) -> None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C diffulab_amd/csrc`).  There is no CPU / PyTorch fallback for the hot path."
            )
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self._fns: dict[str, tuple] = {}
        for name, (restype, argtypes, _) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError here == header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
            self._fns[name] = (fn, restype is ctypes.c_int and name not in _VALUE_RETURNING)

    def call(self, name: str, *args):
        fn, status = self._fns[name]  # (one dict lookup per launch: small-batch steps issue ~700 launches)
        rc = fn(*args)
        if status and rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.cdll.dl_last_error().decode()}")
        return rc


@lru_cache(maxsize=1)
def lib() -> _Lib:
    return _Lib()


def available() -> bool:
    return os.path.exists(LIB_PATH)
