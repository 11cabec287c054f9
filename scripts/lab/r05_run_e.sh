python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "qk_norm_backward_as_its_epilogue" 2>&1 | tail -3
for v in "" qknlab1; do
  if [ -n "$v" ]; then export DIFFULAB_HIP_LIB=$PWD/diffulab_amd/csrc/variants/libdiffulab_hip_$v.so; else unset DIFFULAB_HIP_LIB; fi
  python scripts/lab/attn_bwd_qkn_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
done
unset DIFFULAB_HIP_LIB
for m in 1 0 1 0; do echo "DL_ATTN_BWD_QKN=$m $(DL_ATTN_BWD_QKN=$m python scripts/train_step_bench.py s2 --batch 256 2>&1 | grep -v amdgpu.ids | tail -1)"; done
python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "conv3x3_implicit" 2>&1 | tail -3
for m in 1 0 1 0; do echo "DL_LAB_CONV_BIG=$m $(DL_LAB_CONV_BIG=$m python scripts/unet_bench.py 2>&1 | grep -v amdgpu.ids | tail -1)"; done
