# end of round 6, FINAL tree: bench line + kernel stats + PMC passes, the other configurations' step times, UNet kernel stats / chain at B = 128 and 64, whole GPU suite
export TMPDIR=/tmp
ROOT=$(pwd)
bash scripts/profile_round.sh r06_zz > gpurun_out/r06_zz_profile.log 2>&1; tail -3 gpurun_out/r06_zz_profile.log
{
for c in "cifar 32" "repa 128" "repa_rs 128" "sprint 32" "sprint 256" "ddt 256" "joint 32" "sprint_joint 32" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/unet_bench.py --batch 64 --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --steps 30 --warmup 8 2>&1 | grep workload
python scripts/fp32_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/unet_sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
} > gpurun_out/r06_zz_step_times.txt 2>&1
cat gpurun_out/r06_zz_step_times.txt | cut -c1-200
for b in 128 64; do
OUT=$ROOT/gpurun_out/r06_zz_unet; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --batch $b --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = $b), end of round 6 (final tree)" > gpurun_out/r06_zz_unet_b${b}_kernel_stats.txt
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); python3 scripts/lab/step_chain.py $T > gpurun_out/r06_zz_unet_b${b}_step_chain.txt 2>&1
rm -rf $OUT
done
head -6 gpurun_out/r06_zz_unet_b128_step_chain.txt
timeout 1500 python -m pytest tests -q -x -m gpu --durations=5 > gpurun_out/r06_zz_pytest_gpu.txt 2>&1; tail -8 gpurun_out/r06_zz_pytest_gpu.txt
