# round 6, session r: UNet step chain with the tap-reusing weight gradient as the default (partial images, 128 workgroups)
export TMPDIR=/tmp
ROOT=$(pwd)
for b in 128 64; do
OUT=$ROOT/gpurun_out/r06_r_unet; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --batch $b --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = $b), round 6 session r" > gpurun_out/r06_r_unet_b${b}_kernel_stats.txt
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); python3 scripts/lab/step_chain.py $T > gpurun_out/r06_r_unet_b${b}_step_chain.txt 2>&1
rm -rf $OUT
done
cut -c1-150 gpurun_out/r06_r_unet_b128_step_chain.txt
