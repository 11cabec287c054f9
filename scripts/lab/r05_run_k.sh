python -m pytest tests/test_unet_gpu.py tests/test_unet_fp32_gpu.py tests/test_full_dims_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do python scripts/unet_bench.py 2>&1 | grep -v amdgpu.ids | tail -1; done
