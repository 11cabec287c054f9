# LAB (round 5): A/B of the "de-phased operand issue" builds of the persistent GEMMs (gemm_nt_big_k, gemm_nt_rows_k, mlp_dswiglu_rc_k):
# waves 0-3 issue their share of the next stage's DMA right behind the barrier, their SIMD partners (waves 4-7) after NT_DEPHASE of
# the k-step's four 16-deep sub-steps.  Build:  scripts/lab/build_variant.sh deph2 "-DNT_DEPHASE=2" gemm.hip gemm_ln.hip mlp_bwd.hip
for v in "" deph1 deph2 deph3 ""; do
  if [ -n "$v" ]; then export DIFFULAB_HIP_LIB=$PWD/diffulab_amd/csrc/variants/libdiffulab_hip_$v.so; else unset DIFFULAB_HIP_LIB; fi
  echo "=== variant: ${v:-base}"
  python scripts/gemm_bench.py nt 2>&1 | grep -v amdgpu.ids | tail -12
  python scripts/row_gemm_bench.py 2>&1 | grep -v amdgpu.ids
  python scripts/lab/rc_components.py 2>&1 | grep -v amdgpu.ids
  python scripts/train_step_bench.py s2 --batch 256 2>&1 | grep -v amdgpu.ids | tail -2
done
