# round 6, session f: parity of the edge-shifted grouped weight gradient, the fp8-QK-only lab kernel, launch plans, the B = 256 test with
# its fixture; the joint engines with / without the grouped weight gradient at 640-wide; main-queue busy time of two joint engines
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_abi.py tests/test_ddt_gpu.py -q -x -m gpu > $OUT/pytest_a.txt 2>&1; tail -3 $OUT/pytest_a.txt
timeout 600 python -m pytest tests/test_attn_fp8_gpu.py -q -s -m gpu > $OUT/pytest_fp8.txt 2>&1; grep -E "fp8-QK-only|passed|failed" $OUT/pytest_fp8.txt | cut -c1-300
(time timeout 900 python -m pytest tests/test_parity_bf16_gpu.py -q -s -m gpu -k "b256") > $OUT/pytest_b256.txt 2>&1; grep -E "B=256|oracle leg|passed|failed|real" $OUT/pytest_b256.txt | cut -c1-300
{
for g in 1 0 1 0; do echo "DL_WGRAD_GROUP=$g $(DL_WGRAD_GROUP=$g python scripts/train_step_bench.py ddt_joint --batch 16 2>&1 | grep -v amdgpu.ids | tail -1)"; done
for c in "sprint_joint 32" "joint 32" "ddt 256"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
} > $OUT/joint_step_times.txt 2>&1
cat $OUT/joint_step_times.txt | cut -c1-220
for c in "sprint_joint 32" "ddt_joint 16"; do set -- $c
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$1 -o kt -- python3 $ROOT/scripts/train_step_bench.py $1 --batch $2 --steps 6 --warmup 4 > $OUT/kt_$1.log 2>&1
  cd $ROOT
  T=$(find $OUT/kt_$1 -name "*kernel_trace.csv" | head -1)
  python3 scripts/lab/step_chain.py $T > $OUT/${1}_step_chain.txt 2>&1
  rm -rf $OUT/kt_$1
  head -14 $OUT/${1}_step_chain.txt | cut -c1-110; grep "queue" $OUT/${1}_step_chain.txt
done
