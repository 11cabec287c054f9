# end of round 6, FINAL tree (after the split-convolution tile rule): whole GPU suite, smoke, bench line, UNet steps and traces
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1500 python -m pytest tests -q -x -m gpu --durations=5 > gpurun_out/r06_zz_pytest_gpu.txt 2>&1; tail -8 gpurun_out/r06_zz_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a gpurun_out/r06_zz_pytest_gpu.txt
python bench.py > gpurun_out/r06_zz3_bench_line.json 2> gpurun_out/r06_zz3_bench.err; cut -c1-330 gpurun_out/r06_zz3_bench_line.json
{
python scripts/unet_bench.py --batch 64 --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --batch 64 --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r06_zz3_unet_step_times.txt 2>&1; cut -c1-200 gpurun_out/r06_zz3_unet_step_times.txt
for b in 128 64; do
OUT=$ROOT/gpurun_out/r06_zz_unet; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --batch $b --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = $b), end of round 6 (final tree)" > gpurun_out/r06_zz_unet_b${b}_kernel_stats.txt
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); python3 scripts/lab/step_chain.py $T > gpurun_out/r06_zz_unet_b${b}_step_chain.txt 2>&1
rm -rf $OUT
done
head -3 gpurun_out/r06_zz_unet_b64_step_chain.txt
