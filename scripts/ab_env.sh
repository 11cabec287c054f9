#!/bin/bash
# scratch A/B: headline step under environment switches
for v in "DL_SIDE_WGS=64" "DL_SIDE_WGS=96" "DL_SIDE_WGS=128" "DL_SIDE_WGS=160" "DL_SIDE_WGS=128 DL_GEMM_TN_W4_XCD=0" "DL_SIDE_WGS=96 DL_GEMM_TN_W4_XCD=0"; do
  echo "$v"
  env DL_GEMM_TN_VARIANT=2 $v python scripts/train_step_bench.py s2 --batch 256 --steps 30 --warmup 8 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
