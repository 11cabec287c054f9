# round 6, session w: the weight-gradient kernel's read stream five fragments ahead and across step boundaries
export TMPDIR=/tmp
OUT=gpurun_out/r06_w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -m gpu -k "conv3x3 or unet_launch or partial_images" > $OUT/pytest_conv.txt 2>&1; tail -3 $OUT/pytest_conv.txt
DL_LAB_WGRAD_HALO=2 python scripts/conv_wgrad_bench.py 128 2>&1 | grep -v amdgpu.ids | cut -c1-72 > $OUT/loop_only_b128.txt; cat $OUT/loop_only_b128.txt
for b in 128 64; do python scripts/conv_wgrad_bench.py $b 2>&1 | grep -v amdgpu.ids > $OUT/wgrad_alone_b${b}.txt; cut -c1-165 $OUT/wgrad_alone_b${b}.txt; done
for b in 128 64; do for v in 1 2; do echo "unet B=$b $(python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done > $OUT/step.txt 2>&1; cat $OUT/step.txt
