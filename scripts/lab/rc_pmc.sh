export TMPDIR=/tmp
ROOT=$(pwd)
for c in 1 2 4; do
  export DL_RC_CSPLIT=$c
  OUT=$ROOT/gpurun_out/rc_pmc_$c; mkdir -p $OUT
  cd /tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $ROOT/scripts/mlp_rc_bench.py > $OUT/f.log 2>&1
  cd $ROOT
  F=$(find $OUT/f -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$c" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if r['Counter_Name']=='FETCH_SIZE': acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'rc_k' in k or 'swiglu' in k.lower(): print('csplit',sys.argv[2],k, 'launches',len(v),'fetch MB/launch (x2 KiB corr.)', round(sum(v)/len(v)*1024*2/1e6,1))
PY
done
