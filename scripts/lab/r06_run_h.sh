# round 6, session h: the pipelined attention backward -- parity (bit for bit against the chain form), kernel A/B, step A/B
export TMPDIR=/tmp
OUT=gpurun_out/r06_h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "pipelined or attention" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/attn_bench.txt; cat $OUT/attn_bench.txt
for m in 1 3 1 3; do echo "DL_LAB_ATTN_PIPE=$m $(DL_LAB_ATTN_PIPE=$m timeout 300 python scripts/train_step_bench.py s2 --batch 256 2>&1 | grep -v amdgpu.ids | tail -1)"; done > $OUT/step_ab.txt 2>&1; cat $OUT/step_ab.txt
