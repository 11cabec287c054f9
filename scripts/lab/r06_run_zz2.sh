# end of round 6, final tree: UNet kernel stats / step chains again (the weight-gradient kernel now prints its name) + PMC traffic and MFMA-busy passes of the UNet step
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 600 python -m pytest tests/test_unet_gpu.py -q -x -m gpu -k "conv3x3" 2>&1 | tail -2
for b in 128 64; do
OUT=$ROOT/gpurun_out/r06_zz_unet; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/scripts/unet_bench.py --batch $b --steps 10 --warmup 3 > $OUT/kt.log 2>&1
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 scripts/kernel_stats_summary.py {} 13 "unet_bench.py (B = $b), end of round 6 (final tree)" > gpurun_out/r06_zz_unet_b${b}_kernel_stats.txt
T=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); python3 scripts/lab/step_chain.py $T > gpurun_out/r06_zz_unet_b${b}_step_chain.txt 2>&1
rm -rf $OUT
done
OUT=$ROOT/gpurun_out/r06_zz_unet_pmc; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/scripts/unet_bench.py --batch 128 --steps 2 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/scripts/unet_bench.py --batch 128 --steps 2 --warmup 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 $ROOT/scripts/unet_bench.py --batch 128 --steps 2 --warmup 2 > $OUT/pmc_mfma.log 2>&1
cd $ROOT
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W --json gpurun_out/r06_zz_unet_pmc_traffic.json > gpurun_out/r06_zz_unet_pmc_traffic.txt
M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1); python3 scripts/pmc_mfma_util.py $M > gpurun_out/r06_zz_unet_pmc_mfma_util.txt
rm -rf $OUT
head -25 gpurun_out/r06_zz_unet_pmc_traffic.txt; head -14 gpurun_out/r06_zz_unet_pmc_mfma_util.txt
grep -n "queue 2" -A4 gpurun_out/r06_zz_unet_b128_step_chain.txt
