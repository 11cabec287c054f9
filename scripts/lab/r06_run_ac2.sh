export TMPDIR=/tmp
OUT=gpurun_out/r06_ac; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_full_dims_gpu.py -q -x -m gpu -k "conv3x3 or unet" 2>&1 | tail -2
for b in 64; do for v in 0 1; do echo "== conv alone B=$b split128=$v"; CONV_BENCH_B=$b DL_LAB_CONV_SPLIT128=$v python scripts/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "16x16\|8x8\| 4x4" | cut -c1-120; done; done > $OUT/conv_alone_b64_final.txt 2>&1; grep "16x16 Ci= 128\|16x16 Ci= 256\|==" $OUT/conv_alone_b64_final.txt
{
for b in 64 128 32; do for v in 0 1 0 1; do echo "unet B=$b split128=$v $(DL_LAB_CONV_SPLIT128=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/step2.txt 2>&1; cat $OUT/step2.txt
