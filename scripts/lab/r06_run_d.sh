# round 6, session d: atomics-free convolution weight gradients + launch plans -- parity, then the UNet step at B = 64 / 128
export TMPDIR=/tmp
OUT=gpurun_out/r06_d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_abi.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
{
for b in 64 128; do
  for v in "DL_LAUNCH_PLAN=0 DL_UNET_WGRAD_PARTS=0" "DL_LAUNCH_PLAN=0 DL_UNET_WGRAD_PARTS=1" "DL_LAUNCH_PLAN=1 DL_UNET_WGRAD_PARTS=0" "DL_LAUNCH_PLAN=1 DL_UNET_WGRAD_PARTS=1"; do
    echo "B=$b $v $(env $v timeout 300 python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload)"
  done
done
} > $OUT/unet_ab.txt 2>&1
cat $OUT/unet_ab.txt | cut -c1-260
