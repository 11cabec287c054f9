python -m pytest tests/test_dit_gpu.py -x -q -m gpu -k "qk_norm_on_load_path_native or hipgraph or sampler_loops" 2>&1 | tail -3
python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
UNET_HOST_PROFILE=1 python scripts/unet_bench.py 2>&1 | grep -v amdgpu.ids | head -60
