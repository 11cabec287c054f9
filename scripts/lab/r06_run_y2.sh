export TMPDIR=/tmp
OUT=gpurun_out/r06_y2; mkdir -p $OUT
{
for b in 128 64; do for v in 0 1 0 1; do echo "unet B=$b DL_LAB_NT_DEEP=$v $(DL_LAB_NT_DEEP=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/nt_deep_unet_final.txt 2>&1; cat $OUT/nt_deep_unet_final.txt
