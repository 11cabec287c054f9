# round 6, session ae: the q / kv projections of the UNet's AttentionBlocks (and their data gradients) as one launch (dl_gemm_nt_pair)
export TMPDIR=/tmp
OUT=gpurun_out/r06_ae; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_full_dims_gpu.py -q -x -m gpu -k "two_small or unet or gemm" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
{
for b in 64 128 32; do for v in 0 1 0 1; do echo "unet B=$b DL_UNET_NT_PAIR=$v $(DL_UNET_NT_PAIR=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c58-120)"; done; done
} > $OUT/step.txt 2>&1; cat $OUT/step.txt
