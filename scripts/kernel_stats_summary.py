"""rocprofv3 --kernel-trace --stats --output-format csv  ->  compact text table (the *_kernel_stats.csv it writes).
    python scripts/kernel_stats_summary.py <x_kernel_stats.csv> <steps profiled> ["command line"]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {cmd}")
print(f"# total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step (sum over all streams)")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'pct':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for r in rows:
    if float(r["Percentage"]) < 0.005:
        continue
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['Percentage']):6.2f} "
          f"{float(r['AverageNs']) / 1e3:9.2f} {float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f}")
