# end of round 5, final tree: GPU suite, smoke, bench line + kernel stats + PMC, the other configurations' step times
python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r05_z_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_z_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/profile_round.sh r05_z > gpurun_out/r05_z_profile.log 2>&1; tail -3 gpurun_out/r05_z_profile.log
{
for c in "cifar 32" "repa 128" "repa_rs 128" "sprint 32" "sprint 256" "ddt 256" "joint 32" "sprint_joint 32" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/unet_bench.py --batch 64 --steps 30 --warmup 8 2>&1 | grep workload
python scripts/unet_bench.py --steps 30 --warmup 8 2>&1 | grep workload
python scripts/fp32_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r05_z_step_times.txt 2>&1
cat gpurun_out/r05_z_step_times.txt | cut -c1-200
