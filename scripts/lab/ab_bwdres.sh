# LAB (round 4): attention backward with all four tiles resident against the shipped kernel.  The variant is NOT kept in the product
# sources (slower, DESIGN.md section 6 "Round 4"):  git apply scripts/lab/patches/attn_bwd_resident.patch &&
#   scripts/lab/build_variant.sh bwdres "-DATTN_BWD_RES=1" attention.hip && git checkout diffulab_amd/csrc/attention.hip
V=$PWD/diffulab_amd/csrc/build/libdiffulab_hip_bwdres.so
echo "=== parity of the variant (kernel tests + DiT fixtures)"
DIFFULAB_HIP_LIB=$V python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_parity_bf16_gpu.py -m gpu -q -x -k "attn or attention or fixture or oracle or step" 2>&1 | tail -3
for v in base bwdres base bwdres; do
  if [ $v = bwdres ]; then export DIFFULAB_HIP_LIB=$V; else unset DIFFULAB_HIP_LIB; fi
  echo "=== $v"
  python scripts/attn_bench.py 2>&1 | grep "attn_bwd"
  python scripts/train_step_bench.py s2 --batch 256 --steps 30 2>&1 | tail -1
done
