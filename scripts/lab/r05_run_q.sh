python bench.py --trainer-mode --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>gpurun_out/r05_q.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trainer-mode', d['value'], d['ms_per_step'], d['config'].get('trainer_mode'))"
tail -3 gpurun_out/r05_q.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bare', d['value'], d['ms_per_step'])"
python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
python scripts/sampler_bench.py --batch 8 2>&1 | grep -v amdgpu.ids | tail -1
DL_CFG_PAIR=0 python scripts/sampler_bench.py --batch 8 2>&1 | grep -v amdgpu.ids | tail -1
