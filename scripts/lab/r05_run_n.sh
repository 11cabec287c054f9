python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "groupnorm" 2>&1 | tail -3
for m in 1 2 1 2; do echo "DL_LAB_GN_FUSED=$m $(DL_LAB_GN_FUSED=$m python scripts/unet_bench.py 2>&1 | grep -v amdgpu.ids | grep workload | tail -1 | cut -c1-120)"; done
