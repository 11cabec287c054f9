#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/gn_prof
rm -rf $OUT; mkdir -p $OUT
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/scripts/gn_bench.py > $OUT/kt.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gn_prof/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if "gn_" not in n: continue
    key = (n[-28:], r["Grid_Size_X"], r["Grid_Size_Y"])
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, gx, gy), (c, t) in agg.items():
    print(f"{n:30s} grid {gx:>8s} x {gy:>4s}: {c:4d} launches, avg {t/c:7.1f} us")
PY
find $OUT -name "*kernel_trace.csv" -delete
