# round 6, session q: the UNet step with the tap-reusing weight gradient, atomic and partial-image forms
export TMPDIR=/tmp
OUT=gpurun_out/r06_q; mkdir -p $OUT
run() { echo "B=$1 halo=$2 parts=$3 wgs=$4: $(DL_LAB_WGRAD_HALO=$2 DL_UNET_WGRAD_PARTS=$3 DL_UNET_WGRAD_WGS=$4 python scripts/unet_bench.py --batch $1 --steps 30 --warmup 8 2>&1 | grep workload | cut -c40-175)"; }
{
for b in 128 64; do
  for rep in 1 2; do
    run $b 0 0 0
    run $b 1 0 0
    run $b 1 1 0
    run $b 1 1 128
    run $b 1 1 192
  done
done
} > $OUT/unet_step_ab.txt 2>&1
cat $OUT/unet_step_ab.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -m gpu > $OUT/pytest_unet.txt 2>&1; tail -3 $OUT/pytest_unet.txt
DL_UNET_WGRAD_PARTS=1 timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -m gpu > $OUT/pytest_unet_parts.txt 2>&1; tail -3 $OUT/pytest_unet_parts.txt
