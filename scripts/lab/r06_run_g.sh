# round 6, session g: the whole GPU suite on this tree (durations), smoke
export TMPDIR=/tmp
OUT=gpurun_out/r06_g; mkdir -p $OUT
(time python -X faulthandler -m pytest tests -q -m gpu --durations=25) > $OUT/pytest_gpu.txt 2>&1; tail -40 $OUT/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
