run() { echo "== $*"; env "$@" python scripts/gemm_bench.py nt 20 2>&1 | grep -E "d_xm2|d_xm1|d_a |mlp2"; }
run A=1
run DL_GEMM_NT_ANT=1
run A=1
run DL_GEMM_NT_ANT=1
for i in 1 2 3; do
for v in 0 1; do echo "ANT=$v"; DL_GEMM_NT_ANT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o "\"ms_per_step\": [0-9.]*"; done; done
