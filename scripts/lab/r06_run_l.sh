# round 6, session l: the in-backward optimizer (training/optim.py EarlyStep) -- parity, then the UNet / DiT / joint steps with and without it
export TMPDIR=/tmp
OUT=gpurun_out/r06_l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_full_dims_gpu.py tests/test_trainer_gpu.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -12 $OUT/pytest.txt | cut -c1-200
{
for b in 64 128; do for e in "" "--early-step" "" "--early-step"; do echo "unet B=$b ${e:-plain} $(python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 $e 2>&1 | grep workload | cut -c1-170)"; done; done
for e in "" "--early-step" "" "--early-step"; do echo "s2 B=256 ${e:-plain} $(python scripts/train_step_bench.py s2 --batch 256 $e 2>&1 | grep -v amdgpu.ids | tail -1)"; done
for c in "sprint_joint 32" "ddt_joint 16" "cifar 32"; do set -- $c; for e in "" "--early-step"; do echo "$1 B=$2 ${e:-plain} $(python scripts/train_step_bench.py $1 --batch $2 $e 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
} > $OUT/early_step_ab.txt 2>&1
cat $OUT/early_step_ab.txt | cut -c1-260
