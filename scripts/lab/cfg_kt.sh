# LAB: kernel stats of one train_step_bench configuration:  bash scripts/lab/cfg_kt.sh ddt 256
export TMPDIR=/tmp
CFG=$1; B=$2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_$CFG; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/train_step_bench.py $CFG --batch $B --steps 10 > $OUT/kt.log 2>&1
cd $ROOT
tail -1 $OUT/kt.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 1 "$CFG B=$B" | head -${3:-34}
rm -rf $OUT/kt
