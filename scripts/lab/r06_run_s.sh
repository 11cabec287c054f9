# round 6, session s: atomics-aware split counts of the atomic weight-gradient GEMMs
export TMPDIR=/tmp
OUT=gpurun_out/r06_s; mkdir -p $OUT
python scripts/lab/unet_tn_shapes.py 128 2>&1 | grep -v amdgpu.ids > $OUT/tn_shapes_b128.txt; cut -c1-170 $OUT/tn_shapes_b128.txt
python scripts/lab/unet_tn_shapes.py 64 2>&1 | grep -v amdgpu.ids > $OUT/tn_shapes_b64.txt; tail -1 $OUT/tn_shapes_b64.txt
{
for b in 128 64; do for v in 0 1 0 1; do echo "unet B=$b DL_LAB_TN_SPLIT_MODEL=$v $(DL_LAB_TN_SPLIT_MODEL=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
for c in "cifar 32" "cifar 256" "repa 128" "ddt_joint 16" "sprint_joint 32"; do set -- $c; for v in 0 1 0 1; do echo "$1 B=$2 DL_LAB_TN_SPLIT_MODEL=$v $(DL_LAB_TN_SPLIT_MODEL=$v python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120)"; done; done
} > $OUT/step_ab.txt 2>&1
cat $OUT/step_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
