# round 6, session e: is the UNet at B = 64 host-bound?  (kernel trace: main-queue busy time against the step's wall time, launch
# plans off / on);  the B = 256 autocast fixture;  step times + host share of the joint engines on this tree
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_e; mkdir -p $OUT
for p in 0 1; do
  cd /tmp
  DL_LAUNCH_PLAN=$p rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$p -o kt -- python3 $ROOT/scripts/unet_bench.py --batch 64 --steps 6 --warmup 4 > $OUT/kt$p.log 2>&1
  cd $ROOT
  T=$(find $OUT/kt$p -name "*kernel_trace.csv" | head -1)
  python3 scripts/lab/step_chain.py $T > $OUT/unet_b64_plan${p}_step_chain.txt 2>&1
  rm -rf $OUT/kt$p
  head -4 $OUT/unet_b64_plan${p}_step_chain.txt; tail -1 $OUT/kt$p.log | cut -c1-200
done
(time python tests/golden/make_b256_autocast.py $OUT/dit_b256_autocast.npz) > $OUT/make_b256.txt 2>&1; tail -6 $OUT/make_b256.txt
{
for c in "sprint_joint 32" "joint 32" "ddt 256" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/host_profile.py sprint_joint 32 2>&1 | grep -v amdgpu.ids | head -40
} > $OUT/joint_step_times.txt 2>&1
cat $OUT/joint_step_times.txt | cut -c1-220
