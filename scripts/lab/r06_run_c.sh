# round 6, session c: UNet at the configured batch 64 -- eager against the whole step as one hipGraph; host profile
export TMPDIR=/tmp
OUT=gpurun_out/r06_c; mkdir -p $OUT
{
for b in 64 128; do
  timeout 300 python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 --graph 2>&1 | grep -v amdgpu.ids | tail -3
done
UNET_HOST_PROFILE=1 timeout 300 python scripts/unet_bench.py --batch 64 --steps 10 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -45
} > $OUT/unet_graph.txt 2>&1
cat $OUT/unet_graph.txt | cut -c1-250
