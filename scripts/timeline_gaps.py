"""Idle time on the MAIN stream of a training step: kernel-trace CSV of bench.py -> per-step wall, busy time per queue and the
gaps between consecutive main-stream kernels (launch latency the GPU sits out).   python scripts/timeline_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]) for r in rows))
ad = [k for k in ks if k[2].startswith("adamw_k")]
s, e = ad[-2][1], ad[-1][1]
step = [k for k in ks if k[0] >= s and k[1] <= e]
byq = collections.defaultdict(list)
for k in step: byq[k[3]].append(k)
print(f"last step: wall {(e - s) / 1e6:.2f} ms, {len(step)} launches")
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(k[1] - k[0] for k in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    big = sorted(((g, lst[i][2][:40], lst[i + 1][2][:40]) for i, g in enumerate(gaps) if g > 20000), reverse=True)[:8]
    print(f"queue {q}: {len(lst)} kernels, busy {busy / 1e6:.2f} ms, gaps>0: {len(pos)} totalling {sum(pos) / 1e6:.2f} ms "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us)")
    for g, a, b in big: print(f"     gap {g / 1e3:7.1f} us after {a} before {b}")
