for v in "" pp1 pp2 pp3 ""; do
  if [ -n "$v" ]; then export DIFFULAB_HIP_LIB=$PWD/diffulab_amd/csrc/variants/libdiffulab_hip_$v.so; else unset DIFFULAB_HIP_LIB; fi
  echo "=== variant: ${v:-base}"
  GEMM_BENCH_CHECK=1 python scripts/gemm_bench.py nt 2>&1 | grep -v amdgpu.ids | grep -E "^nt (qkv|mlp1 |d_h|d_xm2|d_xm1|d_a)"
done
