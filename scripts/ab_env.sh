#!/bin/bash
for v in "" "--hi-prio" "" "--hi-prio"; do
  echo "flags: $v"
  python scripts/train_step_bench.py s2 --batch 256 --steps 30 --warmup 8 $v 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
python -c "
import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
