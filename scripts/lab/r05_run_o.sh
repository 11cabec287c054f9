python -m pytest tests/test_unet_gpu.py tests/test_dit_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -3
for m in 1 2; do echo "DL_LAB_GN_FUSED=$m $(DL_LAB_GN_FUSED=$m python scripts/unet_bench.py 2>&1 | grep -v amdgpu.ids | grep workload | tail -1 | cut -c1-160)"; done
python scripts/unet_bench.py --batch 64 2>&1 | grep -v amdgpu.ids | grep workload | tail -1 | cut -c1-160
python scripts/train_step_bench.py s2 --batch 256 2>&1 | grep -v amdgpu.ids | tail -1
python scripts/train_step_bench.py cifar --batch 32 2>&1 | grep -v amdgpu.ids | tail -1
python scripts/train_step_bench.py sprint_joint --batch 32 2>&1 | grep -v amdgpu.ids | tail -1
