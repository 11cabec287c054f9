export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_kt7; rm -rf $OUT; mkdir -p $OUT
for m in 0 1 0 1; do echo "DL_LAB_GN_FUSED=$m"; DL_LAB_GN_FUSED=$m python scripts/unet_bench.py --steps 20 --warmup 5 2>&1 | grep workload; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = 128), round 5, fused GroupNorm forward / two-sum backward" > gpurun_out/r05_s_unet_kernel_stats.txt
python3 scripts/lab/step_chain.py $(find $OUT -name "*kernel_trace.csv" | head -1) > gpurun_out/r05_s_unet_step_chain.txt
head -34 gpurun_out/r05_s_unet_step_chain.txt
