python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r05_i_pytest_gpu.txt 2>&1; tail -15 gpurun_out/r05_i_pytest_gpu.txt
python bench.py > gpurun_out/r05_i_bench.json 2> gpurun_out/r05_i_bench.err; tail -c 3000 gpurun_out/r05_i_bench.json | head -c 1500
