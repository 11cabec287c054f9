export TMPDIR=/tmp
OUT=gpurun_out/r06_aa; mkdir -p $OUT
{
for b in 128 64; do for v in 192 128 64 192 128 64; do echo "unet B=$b halo_min_tiles=$v $(DL_LAB_HALO_MIN_TILES=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/halo_min_tiles.txt 2>&1; cat $OUT/halo_min_tiles.txt
