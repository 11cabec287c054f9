"""ctypes binding of ``libdiffulab_comm.so`` (C ABI in ``include/diffulab_comm.h``): the RCCL communicator of the data-parallel
gradient exchange behind plain C entry points.  Loaded only when asked for (``DIFFULAB_DP_BACKEND=abi`` or
``GradReducer(backend="abi")``); the default exchange goes through ``torch.distributed`` (backend "nccl" = RCCL), which is the
same library underneath."""

from __future__ import annotations

import ctypes
import os
from functools import lru_cache

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffulab_comm.so")


@lru_cache(maxsize=1)
def lib() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C diffulab_amd/csrc`")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.dl_comm_last_error.restype = ctypes.c_char_p
    for name, args in (("dl_comm_unique_id", [ctypes.c_char_p]), ("dl_comm_init", [ctypes.POINTER(vp), ctypes.c_char_p, i32, i32, i32]),
                       ("dl_comm_destroy", [vp]), ("dl_reduce_scatter_allgather_async", [vp, vp, i64, vp]),
                       ("dl_comm_broadcast_async", [vp, vp, i64, i32, vp]), ("dl_comm_after_event", [vp, vp]),
                       ("dl_comm_wait", [vp, vp]), ("dl_comm_rank", [vp]), ("dl_comm_world", [vp])):
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, i32
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().dl_comm_last_error().decode()}")


class Communicator:
    """one RCCL communicator + comm stream for this process; ``rank`` / ``world`` / ``device`` as in torch.distributed.  The
    128-byte rendezvous token is created on rank 0 and shared through ``exchange`` (default: a torch.distributed object
    broadcast over whatever process group is up, e.g. gloo)."""

    def __init__(self, rank: int, world: int, device: int, exchange=None) -> None:
        L = lib()
        token = ctypes.create_string_buffer(128)
        if rank == 0:
            _check(L.dl_comm_unique_id(token), "dl_comm_unique_id")
        raw = bytes(token.raw)
        if world > 1:
            if exchange is None:
                import torch.distributed as dist

                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = exchange(raw)
        self._h = ctypes.c_void_p()
        _check(L.dl_comm_init(ctypes.byref(self._h), raw, rank, world, device), "dl_comm_init")
        self.rank, self.world = rank, world

    def all_reduce_async(self, ptr: int, count: int, after_stream: int) -> None:
        _check(lib().dl_reduce_scatter_allgather_async(self._h, ptr, count, after_stream), "dl_reduce_scatter_allgather_async")

    def broadcast_async(self, ptr: int, count: int, root: int, after_stream: int) -> None:
        _check(lib().dl_comm_broadcast_async(self._h, ptr, count, root, after_stream), "dl_comm_broadcast_async")

    def after_event(self, event: int) -> None:
        _check(lib().dl_comm_after_event(self._h, event), "dl_comm_after_event")

    def wait(self, stream: int) -> None:
        _check(lib().dl_comm_wait(self._h, stream), "dl_comm_wait")

    def close(self) -> None:
        if self._h:
            lib().dl_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
