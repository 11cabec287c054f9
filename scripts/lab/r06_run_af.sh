export TMPDIR=/tmp
OUT=gpurun_out/r06_af; mkdir -p $OUT
timeout 600 python -m pytest tests/test_unet_gpu.py -q -x -m gpu -k "gn or group" 2>&1 | tail -2
for v in 0 1; do echo "== gn alone B=64 narrow=$v"; DL_LAB_GN_NARROW=$v python scripts/gn_bench.py 64 2>&1 | grep -v amdgpu.ids | grep "HW= 1024\|HW=  256" | cut -c1-40,150-230; done > $OUT/gn_alone.txt 2>&1; cat $OUT/gn_alone.txt
{
for b in 64 32; do for v in 0 1 0 1; do echo "unet B=$b gn_narrow=$v $(DL_LAB_GN_NARROW=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/step.txt 2>&1; cat $OUT/step.txt
