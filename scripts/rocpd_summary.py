#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table like `--stats` prints.

    python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db [steps] > profiles/rNN_*.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace, rocpd sqlite; durations in ns -> us/ms)")
print(f"# total kernel time {tot / 1e6:.2f} ms over {steps} profiled steps = {tot / 1e6 / steps:.3f} ms/step")
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
for name, n, t, avg, mn, mx in rows:
    print(f"{name[:72]:72s} {n:7d} {t / 1e6:10.3f} {100 * t / tot:6.2f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}")
