# round 6, session ab: conv3x3_big_k's work items row-tile-fastest where the weights are the larger operand (an XCD's items share weight slices)
export TMPDIR=/tmp
OUT=gpurun_out/r06_ab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_unet_gpu.py -q -x -m gpu -k "conv3x3" 2>&1 | tail -2
for b in 128 64; do for v in 0 1; do echo "== conv alone B=$b wlocal=$v"; CONV_BENCH_B=$b DL_LAB_CONV_WLOCAL=$v python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "8x8\| 4x4" | cut -c1-120; done; done > $OUT/conv_alone.txt 2>&1; cat $OUT/conv_alone.txt
{
for b in 128 64; do for v in 0 1 0 1; do echo "unet B=$b wlocal=$v $(DL_LAB_CONV_WLOCAL=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/step.txt 2>&1; cat $OUT/step.txt
