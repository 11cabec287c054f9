export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_c5; rm -rf $OUT; mkdir -p $OUT
python scripts/train_step_bench.py sprint_joint --batch 32 2>&1 | tail -1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/train_step_bench.py sprint_joint --batch 32 --steps 10 > $OUT/kt.log 2>&1
cd $ROOT
tail -1 $OUT/kt.log
python3 scripts/lab/step_chain.py $(find $OUT -name "*kernel_trace.csv" | head -1) > gpurun_out/r05_v_cfg5_step_chain.txt
head -70 gpurun_out/r05_v_cfg5_step_chain.txt
