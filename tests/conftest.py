import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


import faulthandler  # noqa: E402

# a host-side crash of the python process (one unreproduced segfault of the GPU suite in round 3) leaves the Python stack of every
# thread in the driver's pytest.log instead of a bare "Segmentation fault"
faulthandler.enable(all_threads=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so `-m "not gpu"` and a bare run both work on CPU."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name: str):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def probe_lib():
    """libdiffulab_probe.so (include/diffulab_probe.h): LAB instrumentation, not part of the product ABI -- the GPU tests use its two
    instruction-semantics probes to pin the operand layouts the product kernels rely on"""
    import ctypes

    path = os.path.join(ROOT, "diffulab_amd", "libdiffulab_probe.so")
    if not os.path.exists(path):
        pytest.fail(f"{path} not found: `make -C diffulab_amd/csrc probe`")
    lib = ctypes.CDLL(path)
    for name in ("dl_probe_tr16", "dl_probe_mfma_f8"):
        getattr(lib, name).restype = ctypes.c_int
    lib.dl_probe_tr16.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.dl_probe_mfma_f8.argtypes = [ctypes.c_void_p] * 4
    return lib
