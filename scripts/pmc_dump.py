"""Sum rocprofv3 PMC counters per kernel (per launch average): python scripts/pmc_dump.py <csv> [<csv> ...] [--match attn]"""
import csv, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
args = [a for a in args if a != match]
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for path in args:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:40]
            if match and match not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v / n[k][c]:16.0f} per launch  ({n[k][c]} launches)")
