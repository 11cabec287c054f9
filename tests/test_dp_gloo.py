"""Data-parallel reducer on 2 CPU ranks (gloo): the N>1 path of bench.py / the trainer without a GPU.

Checks the contract of SURVEY.md §8e: after `GradReducer` the flat gradient arena holds the SUM over ranks (the
1/world average is applied by the optimizer kernel), ranges may arrive in any block order, buckets are contiguous,
`sync=False` (gradient-accumulation micro-step) reduces nothing, and `broadcast_arena` makes ranks start equal."""

import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank: int, world: int, port: int, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffulab_amd.training.dp import GradReducer, broadcast_arena

    n = 1000
    params = torch.full((n,), float(rank + 1))
    broadcast_arena(params)
    ok = bool((params == 1.0).all())
    grads = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(grads, bucket_bytes=4 * 300)  # ~300-element buckets -> several collectives
    ranges = [(700, 1000), (400, 700), (100, 400)]   # blocks finish last-to-first
    for lo, hi in ranges:
        red.ready(lo, hi)
    red.ready(0, 100)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok &= bool(torch.equal(grads, expect)) and abs(red.grad_scale - 1.0 / world) < 1e-12
    # a block's own range and its slice of the stacked adaLN matrix sit in different parts of the arena: the pending ranges of one
    # flush form several contiguous runs, each reduced on its own; flush=True reduces without waiting for a full bucket
    g3 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red3 = GradReducer(g3, bucket_bytes=4 * 10_000)  # never fills: every reduction below is an explicit flush / finish
    red3.ready(60, 80)        # adaLN rows of the last block
    red3.ready(800, 1000)     # the last block
    ok &= len(red3._pending) == 2 and not red3._works
    red3.ready(40, 60)
    red3.ready(600, 800, flush=True)
    ok &= not red3._pending and len(red3._works) == 2  # runs [40, 80) and [600, 1000)
    red3.ready(0, 40)
    red3.ready(80, 600)
    red3.finish()
    ok &= bool(torch.equal(g3, expect))
    # accumulation micro-step: no communication, gradients untouched
    g2 = torch.ones(n) * (rank + 1)
    red2 = GradReducer(g2)
    red2.sync = False
    red2.ready(0, n)
    red2.finish()
    ok &= bool((g2 == rank + 1).all())
    # overlapped buckets or one exchange after the backward is decided by measurement (DIFFULAB_DP_OVERLAP=auto): synchronising steps
    # 2-7 run overlapped, 8-13 with everything reduced in finish() (median of the first-range-to-finish interval per mode), then every
    # rank keeps the same mode (MAX over ranks; the overlapped default is only left for a > 3 % win of the single exchange);
    # the sums are right in every step of either mode
    g4 = torch.zeros(n)
    red4 = GradReducer(g4, bucket_bytes=4 * 300)
    modes = []
    for it in range(16):
        g4.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1) + it)
        for lo, hi in ranges:
            red4.ready(lo, hi, flush=(lo == 100))
        modes.append((red4.overlap, len(red4._works) > 0))
        red4.ready(0, 100)
        red4.finish()
        ok &= bool(torch.equal(g4, expect + world * it))
    ok &= all(o and wk for o, wk in modes[:8]) and all((not o) and (not wk) for o, wk in modes[8:14])  # (works in flight <=> overlapped)
    ok &= red4.tuned is not None and red4.tuned["mode"] in ("overlapped", "after_backward") and red4._tune is None
    ok &= modes[14][0] == (red4.tuned["mode"] == "overlapped") == modes[15][0]
    out[f"mode{rank}"] = red4.tuned["mode"] if red4.tuned else None
    os.environ["DIFFULAB_DP_OVERLAP"] = "0"  # pinned: nothing is reduced before finish(), not even on flush=True
    # (a pin is a pin from the first step on: no warm-up measurement of the other mode unless DIFFULAB_DP_MEASURE=1 asks for it)
    g5 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red5 = GradReducer(g5, bucket_bytes=4 * 300)
    red5.ready(400, 1000, flush=True)
    ok &= (not red5._works) and len(red5._pending) == 1 and red5._tune is None
    red5.ready(0, 400)
    red5.finish()
    ok &= bool(torch.equal(g5, expect))
    # pinned WITH the measurement (opt-in): both schedules are timed during the same warm-up steps and reported, the pin decides
    os.environ["DIFFULAB_DP_MEASURE"] = "1"
    g6 = torch.zeros(n)
    red6 = GradReducer(g6, bucket_bytes=4 * 300)
    for it in range(16):
        g6.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1) + it)
        for lo, hi in ranges:
            red6.ready(lo, hi, flush=(lo == 100))
        red6.ready(0, 100)
        red6.finish()
        ok &= bool(torch.equal(g6, expect + world * it))
    ok &= red6.tuned is not None and red6.tuned["mode"] == "after_backward" and not red6.overlap and "pinned" in red6.tuned["decided"]
    ok &= red6.tuned["overlapped_ms_per_step"] > 0 and red6.tuned["after_backward_ms_per_step"] > 0
    os.environ.pop("DIFFULAB_DP_OVERLAP")
    os.environ.pop("DIFFULAB_DP_MEASURE")
    out[rank] = ok
    dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)
    assert out["mode0"] == out["mode1"] and out["mode0"] is not None
