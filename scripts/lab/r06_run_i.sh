# round 6, session i: where the pipelined attention forward starts to pay (items per CU = B * 6 / 256)
export TMPDIR=/tmp
OUT=gpurun_out/r06_i; mkdir -p $OUT
python - > $OUT/attn_fwd_gate.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from diffulab_amd import ops
from diffulab_amd.engine import rope_grid_tables
dev, bf = "cuda", torch.bfloat16
H, N, dh = 6, 256, 64
D = H * dh
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cos, sin = (z.to(dev) for z in rope_grid_tables(16, 16, [32, 32], 10_000.0))
sq, sk = torch.ones(D, device=dev), torch.ones(D, device=dev)
for B in (128, 144, 171, 192, 214, 256, 384, 512):
    qkv = torch.randn(B * N, 3 * D, device=dev).to(bf)
    ssq = (torch.rand(B * N, 2, device=dev) + 0.5) * D
    q, k = (torch.empty(B, H, N, dh, device=dev, dtype=bf) for _ in range(2))
    rr = torch.empty(B * N, 2, device=dev)
    out, lse = torch.empty(B, N, D, device=dev, dtype=bf), torch.empty(B, H, N, device=dev)
    row = []
    for mode in (0, 1, 0, 1):
        ops.lib().cdll.dl_lab_set_attn_pipe(mode)
        t = timeit(lambda: ops.attn_fwd_qkn(qkv, ssq, sq, sk, cos, sin, q, k, rr, out, lse, B, H, N, dh, 64, dh ** -0.5))
        ti = timeit(lambda: ops.attn_fwd_qkn(qkv, ssq, sq, sk, cos, sin, None, None, None, out, lse, B, H, N, dh, 64, dh ** -0.5))
        row.append((mode, t, ti))
    print(f"B={B:4d} items/CU={B * H / 256:5.2f}  " + "  ".join(f"pipe={m}: train {t:6.1f} infer {ti:6.1f}" for m, t, ti in row))
PY
cat $OUT/attn_fwd_gate.txt | grep -v amdgpu
