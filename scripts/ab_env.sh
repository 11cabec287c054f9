#!/bin/bash
# scratch A/B
for v in "DL_GEMM_NT_PIPE=0" "DL_GEMM_NT_PIPE=1"; do
  echo "$v"
  env $v python scripts/gemm_bench.py nt 2>/dev/null | grep -v amdgpu
done
for v in "DL_GEMM_NT_PIPE=0" "DL_GEMM_NT_PIPE=1" "DL_GEMM_NT_PIPE=0" "DL_GEMM_NT_PIPE=1"; do
  echo "$v"
  env $v python scripts/train_step_bench.py s2 --batch 256 --steps 30 --warmup 8 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
