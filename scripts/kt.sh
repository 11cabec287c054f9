#!/bin/bash
# kernel-trace stats of the bench step:  bash scripts/kt.sh <tag> [ENV=VAL ...]  -> gpurun_out/<tag>/kernel_stats.txt
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 7 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline ($*; 7 steps in the trace)" > $OUT/kernel_stats.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
head -24 $OUT/kernel_stats.txt
