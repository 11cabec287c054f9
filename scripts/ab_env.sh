#!/bin/bash
for v in "DL_GEMM_NT_PIPE=0" "DL_GEMM_NT_PIPE=1"; do
  echo "$v"
  for c in "joint 16" "sprint_joint 64" "sprint 256" "ddt 256" "repa 128" "cifar 32"; do
    set -- $c
    env $v python scripts/train_step_bench.py $1 --batch $2 --steps 20 --warmup 6 2>/dev/null | grep -o '"config": "[a-z_]*", "batch": [0-9]*\|"ms_per_step": [0-9.]*' | paste - -
  done
done
