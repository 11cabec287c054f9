export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_kt5; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench (conv3x3_big_k on)" > gpurun_out/r05_f_unet_kernel_stats.txt
head -50 gpurun_out/r05_f_unet_kernel_stats.txt
