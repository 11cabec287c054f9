"""LAB: last training step of a kernel-trace CSV of bench.py: per queue, time by kernel name (sum, count, mean) in launch order of
first appearance, and the wall-clock cover of each queue.   python scripts/lab/step_chain.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r["Queue_Id"]) for r in rows))
ad = [k for k in ks if k[2].startswith("adamw_k")]
s, e = ad[-2][1], ad[-1][1]
step = [k for k in ks if k[0] >= s and k[1] <= e]
print(f"last step: wall {(e - s) / 1e6:.3f} ms, {len(step)} launches")
byq = collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    agg = collections.OrderedDict()
    for a, b, n, _ in lst:
        t = agg.setdefault(n, [0, 0]); t[0] += b - a; t[1] += 1
    print(f"== queue {q}: {len(lst)} kernels, busy {sum(b - a for a, b, _, _ in lst) / 1e6:.3f} ms")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"   {t / 1e6:7.3f} ms  {c:4d} x {t / c / 1e3:7.1f} us  {n}")
if len(sys.argv) > 2:  # the launch sequence of the step, small kernels only (name, queue, start offset us, duration us)
    for a, b, n, q in step:
        if b - a < int(sys.argv[2]) * 1000: print(f"   +{(a - s) / 1e3:9.1f} us  q{q}  {(b - a) / 1e3:7.1f} us  {n}")
