export TMPDIR=/tmp
OUT=gpurun_out/r06_x; mkdir -p $OUT
{
for b in 64 128; do for rep in 1 2; do
 echo "unet B=$b plain $(python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"
 echo "unet B=$b --early-step $(python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 --early-step 2>&1 | grep workload | cut -c58-120)"
done; done
} > $OUT/early_step.txt 2>&1; cat $OUT/early_step.txt
