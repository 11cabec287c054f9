export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_kt8; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --steps 6 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
ls -la $OUT/kt | head
python3 - <<'PY'
import csv, glob, collections
out = glob.glob("gpurun_out/unet_kt8/kt/*")
kt = [f for f in out if f.endswith("kernel_trace.csv")][0]
ha = [f for f in out if "hip_api_trace" in f][0]
K = list(csv.DictReader(open(kt)))
H = {r["Correlation_Id"]: r for r in csv.DictReader(open(ha))}
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:40], r["Correlation_Id"]) for r in K)
qs = collections.Counter(e[2] for e in ev)
mainq = max(qs, key=qs.get)
main = [e for e in ev if e[2] == mainq]
ad = [i for i, x in enumerate(main) if x[3].startswith("adamw_k")]
step = main[ad[-2] + 1: ad[-1] + 1]
t00 = step[0][0]
rows = []
for i in range(len(step) - 1):
    g = step[i + 1][0] - step[i][1]
    h = H.get(step[i + 1][4])
    if h is None:
        continue
    rows.append((g, step[i][3], step[i + 1][3], (int(h["Start_Timestamp"]) - step[i][1]) / 1e3, (step[i + 1][0] - int(h["End_Timestamp"])) / 1e3))
rows.sort(reverse=True)
print("largest main-queue gaps: gap us | prev kernel | next kernel | host launch call START relative to prev kernel END (us; > 0 = host late) | kernel start after launch call returned (us)")
for g, a, b, late, lat in rows[:25]:
    print(f"{g/1e3:8.1f} | {a:40s} | {b:40s} | {late:9.1f} | {lat:8.1f}")
late_sum = sum(g for g, a, b, late, lat in rows if late > 0 and g > 15000)
print("sum of gaps > 15 us:", sum(g for g, *_ in rows if g > 15000) / 1e6, "ms; of which the host issued the next launch AFTER the previous kernel had ended:", late_sum / 1e6, "ms")
PY
