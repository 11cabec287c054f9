export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_kt10; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = 128), end of round 5 (final tree)" > gpurun_out/r05_z_unet_kernel_stats.txt
python3 scripts/lab/step_chain.py $(find $OUT -name "*kernel_trace.csv" | head -1) > gpurun_out/r05_z_unet_step_chain.txt
cat gpurun_out/r05_z_unet_step_chain.txt | head -80
