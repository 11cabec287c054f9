python -m pytest tests/test_trainer_gpu.py tests/test_dropin.py tests/test_fp32_gpu.py tests/test_unet_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
python bench.py --trainer-mode --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-300
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
