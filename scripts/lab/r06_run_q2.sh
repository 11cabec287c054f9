export TMPDIR=/tmp
OUT=gpurun_out/r06_q; mkdir -p $OUT
run() { echo "B=$1 halo=$2 parts=$3 wgs=$4: $(DL_LAB_WGRAD_HALO=$2 DL_UNET_WGRAD_PARTS=$3 DL_UNET_WGRAD_WGS=$4 python scripts/unet_bench.py --batch $1 --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; }
{
for b in 128 64; do
  for w in 48 64 96 128 160; do run $b 1 1 $w; done
  for w in 64 96 128; do run $b 1 0 $w; done
done
} > $OUT/unet_step_wgs_sweep.txt 2>&1
cat $OUT/unet_step_wgs_sweep.txt
