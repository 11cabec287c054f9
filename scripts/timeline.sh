#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
cd $R
python3 scripts/timeline_gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1)
find $OUT -name "*kernel_trace.csv" -delete
