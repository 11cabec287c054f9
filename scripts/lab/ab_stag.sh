# LAB (round 4): A/B of "staggered start" builds of the attention kernels (-DATTN_STAG=<iterations of s_sleep 127> -DATTN_STAG_SHIFT=<bit of
# blockIdx that picks the delayed half>), scripts/lab/build_variant.sh stagI_S "-DATTN_STAG=I -DATTN_STAG_SHIFT=S" attention.hip
for v in "" stag2_8 stag3_8 stag4_8 stag3_0 stag3_5; do
  if [ -n "$v" ]; then export DIFFULAB_HIP_LIB=$PWD/diffulab_amd/csrc/build/libdiffulab_hip_$v.so; else unset DIFFULAB_HIP_LIB; fi
  echo "=== variant: ${v:-base}"
  python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
