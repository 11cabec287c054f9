# LAB (round 4): A/B of "rotated k-walk" builds of the persistent NT GEMMs.  The variant is a two-line patch that is NOT kept in the
# product sources (measured +-0, DESIGN.md section 6 "Round 4"): in gemm_nt_big_k / gemm_nt_rows_k `stage_next` stages k-step
# (s_kt + (NT_KROT * (blockIdx.x >> 3)) % nk) % nk instead of s_kt; build with
#   scripts/lab/build_variant.sh krot1 "-DNT_KROT=1" gemm.hip gemm_ln.hip
for v in "" krot1 krot5; do
  if [ -n "$v" ]; then export DIFFULAB_HIP_LIB=$PWD/diffulab_amd/csrc/build/libdiffulab_hip_$v.so; else unset DIFFULAB_HIP_LIB; fi
  echo "=== variant: ${v:-base}"
  python scripts/row_gemm_bench.py 2>&1 | grep -v amdgpu.ids
  python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -14
  python scripts/train_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
done
