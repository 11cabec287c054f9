# round 6, final tree: the whole GPU suite + smoke (record for profiles/)
export TMPDIR=/tmp
(time python -X faulthandler -m pytest tests -q -m gpu --durations=15) > gpurun_out/r06_z_pytest_gpu.txt 2>&1; tail -22 gpurun_out/r06_z_pytest_gpu.txt | cut -c1-180
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
