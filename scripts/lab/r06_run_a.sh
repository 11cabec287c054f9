# round 6, session a: where the fills of a steady-state step come from; the full launch sequence of one step; suite durations
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_a; mkdir -p $OUT
python scripts/find_fills.py > $OUT/fills.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/kt.log 2>&1
cd $ROOT
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 scripts/lab/step_chain.py $T 100000 > $OUT/step_chain_full.txt 2>&1
rm -rf $OUT/kt
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -m pytest tests -q -m gpu --durations=60 > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
head -40 $OUT/fills.txt
