# end of round 6, final tree: bench line + kernel stats + PMC passes, the other configurations' step times, UNet kernel stats / chain at B = 128
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 300 python -m pytest tests/test_row_gemm_gpu.py -q -m gpu -k pipelined 2>&1 | tail -2
bash scripts/profile_round.sh r06_z > gpurun_out/r06_z_profile.log 2>&1; tail -3 gpurun_out/r06_z_profile.log
{
for c in "cifar 32" "repa 128" "repa_rs 128" "sprint 32" "sprint 256" "ddt 256" "joint 32" "sprint_joint 32" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/unet_bench.py --batch 64 --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --steps 30 --warmup 8 2>&1 | grep workload
python scripts/fp32_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r06_z_step_times.txt 2>&1
cat gpurun_out/r06_z_step_times.txt | cut -c1-200
OUT=$ROOT/gpurun_out/r06_z_unet; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = 128), end of round 6 (final tree)" > gpurun_out/r06_z_unet_kernel_stats.txt
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); python3 scripts/lab/step_chain.py $T > gpurun_out/r06_z_unet_step_chain.txt 2>&1
rm -rf $OUT
head -8 gpurun_out/r06_z_unet_step_chain.txt
