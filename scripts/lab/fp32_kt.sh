export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_fp32_${2:-unet}; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/fp32_step_bench.py ${1:-64} 3 ${2:-unet} > $OUT/kt.log 2>&1
cd $ROOT
tail -1 $OUT/kt.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 1 "${2:-unet} fp32 B=${1:-64}" | head -24
rm -rf $OUT/kt
