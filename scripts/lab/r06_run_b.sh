# round 6, session b: pipelined attention forward -- parity (bit for bit against the chain form), kernel A/B, step A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06_b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_row_gemm_gpu.py tests/test_dit_gpu.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/attn_bench.txt; cat $OUT/attn_bench.txt
for m in 0 1 0 1; do echo "DL_LAB_ATTN_PIPE=$m $(DL_LAB_ATTN_PIPE=$m timeout 300 python scripts/train_step_bench.py s2 --batch 256 2>&1 | grep -v amdgpu.ids | tail -1)"; done > $OUT/step_ab.txt 2>&1; cat $OUT/step_ab.txt
timeout 300 python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/sampler.txt; cat $OUT/sampler.txt
