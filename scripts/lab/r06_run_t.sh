# round 6, session t: four-slot ring of the 128 x 128 weight-gradient kernel for launches of at most one workgroup per CU
export TMPDIR=/tmp
OUT=gpurun_out/r06_t; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "atomic_weight_gradient or gemm_tn" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python scripts/lab/unet_tn_shapes.py 128 2>&1 | grep -v amdgpu.ids > $OUT/tn_shapes_b128.txt; cut -c1-200 $OUT/tn_shapes_b128.txt
python scripts/lab/unet_tn_shapes.py 64 2>&1 | grep -v amdgpu.ids > $OUT/tn_shapes_b64.txt; tail -1 $OUT/tn_shapes_b64.txt
{
for b in 128 64; do for v in 0 1 0 1; do echo "unet B=$b DL_LAB_TN_DEEP=$v $(DL_LAB_TN_DEEP=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/step_ab.txt 2>&1
cat $OUT/step_ab.txt
