#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/joint_prof
mkdir -p $OUT
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/scripts/train_step_bench.py joint --batch 16 --steps 8 --warmup 4 > $OUT/kt.log 2>&1
cd $R
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 12 "joint B=16, 12 steps in trace" | head -30
find $OUT -name "*kernel_trace.csv" -delete
