# round 6, session u: bias / residual of the 128 x 128 GEMM kernel's epilogue requested ahead of their use
export TMPDIR=/tmp
OUT=gpurun_out/r06_u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_row_gemm_gpu.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python scripts/lab/unet_nt_shapes.py 64 2>&1 | grep -v amdgpu.ids > $OUT/nt_shapes_b64_after.txt; cut -c1-150 $OUT/nt_shapes_b64_after.txt
python scripts/lab/unet_nt_shapes.py 128 2>&1 | grep -v amdgpu.ids > $OUT/nt_shapes_b128_after.txt; tail -1 $OUT/nt_shapes_b128_after.txt
{
for b in 128 64; do for v in 1 2; do echo "unet B=$b $(python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
for c in "cifar 32" "sprint 32" "ddt_joint 16"; do set -- $c; for v in 1 2; do echo "$1 B=$2 $(python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120)"; done; done
} > $OUT/step.txt 2>&1
cat $OUT/step.txt
