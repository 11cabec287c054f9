# round 6, session v: whole GPU suite + bench line on the tree with the tap-reusing weight gradient, the split model and the epilogue prefetch
export TMPDIR=/tmp
OUT=gpurun_out/r06_v; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -x -m gpu --durations=8 > $OUT/pytest_gpu.txt 2>&1; tail -14 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
python bench.py --no-cpu-baseline > $OUT/bench_line2.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_line2.json
