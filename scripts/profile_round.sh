#!/bin/bash
# end-of-round evidence on the GPU box: bench line, rocprofv3 kernel stats of the same step, PMC traffic passes (separate runs).
#   bash scripts/profile_round.sh <tag>        -> gpurun_out/<tag>/{bench_line.json,kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,...}
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_mfma.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 7 "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline (7 steps in the trace, model build included)" > $OUT/kernel_stats.txt
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W --json $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
# keep the merged payload small: the raw traces are not needed
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1); python3 scripts/pmc_mfma_util.py $M > $OUT/pmc_mfma_util.txt
head -12 $OUT/kernel_stats.txt; head -12 $OUT/pmc_traffic.txt; tail -c 600 $OUT/bench_line.json
