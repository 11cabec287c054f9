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


def _host_threads() -> int:
    """cores this process may really use (affinity mask, cgroup cpu quota), capped at 32 -- bench.py's rule for its CPU baseline"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle legs are most of the GPU suite's wall time, and torch sizes its thread pool by the host's LOGICAL cpu count: on a
    # GPU box whose cgroup grants a fraction of a 256-thread host the oversubscribed pool is several times slower than a right-sized
    # one (round 6: the B = 256 oracle leg ran at 2.2 images/s inside pytest against 15.8 images/s in bench.py's cpu_baseline)
    import torch

    torch.set_num_threads(_host_threads())


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
