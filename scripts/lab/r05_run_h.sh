for c in "sprint_joint 32" "joint 32" "ddt_joint 16"; do set -- $c; python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1; done
python scripts/host_profile.py sprint_joint 32 2>&1 | grep -v amdgpu.ids | head -60
python scripts/lab/attn_bwd_qkn_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
