#!/usr/bin/env python
"""per-step view of a rocpd kernel trace of bench.py: wall, GPU-busy union, concurrency, per-kernel totals of the LAST step"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
ad = [r for r in rows if r[0].startswith("adamw_k")]
s, e = ad[-2][2], ad[-1][2]
step = [r for r in rows if r[1] >= s and r[2] <= e]
tot = sum(r[2] - r[1] for r in step)
ev = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
busy = over = depth = 0
last = None
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: over += t - last
    depth += d; last = t
print(f"last step: wall {(e-s)/1e6:.2f} ms | sum of kernel durations {tot/1e6:.2f} ms | GPU busy (union) {busy/1e6:.2f} ms | >=2 kernels in flight {over/1e6:.2f} ms | idle {((e-s)-busy)/1e6:.2f} ms | {len(step)} launches")
c = collections.Counter(); n = collections.Counter()
for r in step: c[r[0][:64]] += r[2] - r[1]; n[r[0][:64]] += 1
for k, v in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 16): print(f"{v/1e6:7.3f} ms  x{n[k]:<4d} {k}")
