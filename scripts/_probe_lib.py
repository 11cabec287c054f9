"""loader of libdiffulab_probe.so (include/diffulab_probe.h): LAB instrumentation for the *_probe.py scripts, not part of the
product ABI and never loaded by the package (`make -C diffulab_amd/csrc probe`)"""
import ctypes
import os

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffulab_amd", "libdiffulab_probe.so")


class _Probe:
    def __init__(self) -> None:
        self.cdll = ctypes.CDLL(_PATH)
        self.cdll.dl_probe_last_error.restype = ctypes.c_char_p
        v, i, q = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        self.cdll.dl_probe_mfma.argtypes = [i, i, v, v, v]
        self.cdll.dl_probe_dma.argtypes = [i, v, v, q, q, v, v]

    def call(self, name: str, *args) -> None:
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.cdll.dl_probe_last_error().decode()}")


def lib() -> _Probe:
    return _Probe()
