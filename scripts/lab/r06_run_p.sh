# round 6, session p: tap-reusing convolution weight gradient (conv_wgrad.hip)
export TMPDIR=/tmp
OUT=gpurun_out/r06_p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -m gpu -k "conv3x3" > $OUT/pytest_conv.txt 2>&1; tail -5 $OUT/pytest_conv.txt
for b in 128 64; do
  DL_LAB_WGRAD_HALO=1 python scripts/conv_wgrad_bench.py $b 2>&1 | grep -v amdgpu.ids > $OUT/wgrad_alone_b${b}_halo.txt
  DL_LAB_WGRAD_HALO=0 python scripts/conv_wgrad_bench.py $b 2>&1 | grep -v amdgpu.ids > $OUT/wgrad_alone_b${b}_old.txt
  cut -c1-160 $OUT/wgrad_alone_b${b}_halo.txt; cut -c1-160 $OUT/wgrad_alone_b${b}_old.txt
done
