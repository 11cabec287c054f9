export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_kt6; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = 128), end of round 5" > gpurun_out/r05_r_unet_kernel_stats.txt
python3 scripts/lab/step_chain.py $(find $OUT -name "*kernel_trace.csv" | head -1) > gpurun_out/r05_r_unet_step_chain.txt
head -30 gpurun_out/r05_r_unet_step_chain.txt
for c in "cifar 32" "repa 128" "repa_rs 128" "sprint 32" "sprint 256" "ddt 256" "joint 32" "sprint_joint 32" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/unet_bench.py --batch 64 2>&1 | grep workload
python scripts/unet_bench.py 2>&1 | grep workload
python scripts/fp32_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
