export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/unet_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/scripts/unet_bench.py --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/scripts/unet_bench.py --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $ROOT
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W --json $OUT/pmc_traffic.json > gpurun_out/r05_u_unet_pmc_traffic.txt
find $OUT -name "*.csv" -size +8M -delete
head -40 gpurun_out/r05_u_unet_pmc_traffic.txt
python scripts/lab/generate_graph_consistency.py 2>&1 | grep -v amdgpu | tail -14
