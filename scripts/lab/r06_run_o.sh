# round 6, session o: the four-slot ring of the 128 x 128 GEMM kernel for launches of at most one workgroup per CU
export TMPDIR=/tmp
OUT=gpurun_out/r06_o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -x -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
python - > $OUT/gemm_ab.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from diffulab_amd import ops
dev, bf = "cuda", torch.bfloat16
def timeit(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in [(8192, 512, 512), (4096, 512, 512), (4096, 1536, 512), (4096, 1024, 512), (2048, 1024, 1024), (2048, 3072, 1024), (1024, 1024, 1024), (1024, 3072, 1024), (8192, 256, 256), (2048, 512, 2048), (16384, 640, 640), (2048, 640, 640), (2048, 1920, 640), (2048, 5120, 640)]:
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf); o = torch.empty(M, N, device=dev, dtype=bf)
    row = []
    for mode in (0, 1, 0, 1):
        ops.lib().cdll.dl_lab_set_nt_deep(mode)
        row.append(timeit(lambda: ops.gemm_nt(a, w, o)))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d} tiles128={(M + 127) // 128 * ((N + 127) // 128):5d}  two-slot {row[0]:6.1f} {row[2]:6.1f} us   four-slot {row[1]:6.1f} {row[3]:6.1f} us   ({fl / row[3] / 1e6:6.1f} TF/s)")
PY
cat $OUT/gemm_ab.txt | grep -v amdgpu
{
for b in 64 128; do for v in 0 1 0 1; do echo "unet B=$b DL_LAB_NT_DEEP=$v $(DL_LAB_NT_DEEP=$v python scripts/unet_bench.py --batch $b --steps 30 --warmup 8 2>&1 | grep workload | cut -c1-150)"; done; done
for c in "cifar 32" "sprint 32" "ddt_joint 16" "sprint_joint 32"; do set -- $c; for v in 0 1 0 1; do echo "$1 B=$2 DL_LAB_NT_DEEP=$v $(DL_LAB_NT_DEEP=$v python scripts/train_step_bench.py $1 --batch $2 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
} > $OUT/step_ab.txt 2>&1
cat $OUT/step_ab.txt | cut -c1-230
