"""micro-benchmark of the HBM-bound row kernels at the DiT-S/2 (B=256) shape: [65536, 384] bf16 rows.
    python scripts/row_kernel_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops

dev = "cuda"
B, N, D = 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 384   # python scripts/row_kernel_bench.py [D]
if D > 384:
    B = 128
M = B * N
bf = torch.bfloat16
x, t = torch.randn(M, D, device=dev).to(bf), torch.randn(M, D, device=dev).to(bf)
mod = torch.randn(B, 6 * D, device=dev).to(bf)
w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
out, xo = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


us = timeit(lambda: ops.ln_modulate_fwd(x, w, b, mod[:, :D], mod[:, D:2 * D], N, 1e-5, out, mean, rstd))
print(f"ln_mod_fwd plain          : {us:7.1f} us  {2 * M * D * 2 / us / 1e6:5.2f} TB/s")
us = timeit(lambda: ops.ln_modulate_fwd(x, w, b, mod[:, :D], mod[:, D:2 * D], N, 1e-5, out, mean, rstd, t=t,
                                        gate=mod[:, 2 * D:3 * D], x_out=xo))
print(f"ln_mod_fwd + gated resid  : {us:7.1f} us  {4 * M * D * 2 / us / 1e6:5.2f} TB/s")

# backward (+ the fused backward of the gated residual that follows): dout, x, dres, t in; dx, dt out; f32 accumulators by atomics
dout, dres = torch.randn(M, D, device=dev).to(bf), torch.randn(M, D, device=dev).to(bf)
ops.ln_modulate_fwd(x, w, b, mod[:, :D], mod[:, D:2 * D], N, 1e-5, out, mean, rstd)
dx, dt = torch.empty_like(x), torch.empty_like(x)
dmod = torch.zeros(B, 6 * D, device=dev)
dwb = torch.zeros(B, 2, D, device=dev)
us = timeit(lambda: ops.ln_modulate_bwd(dout, x, w, b, mod[:, :D], N, mean, rstd, dres, dx, dmod[:, :D], dmod[:, D:2 * D], dwb))
print(f"ln_mod_bwd plain          : {us:7.1f} us  {4 * M * D * 2 / us / 1e6:5.2f} TB/s")
us = timeit(lambda: ops.ln_modulate_bwd(dout, x, w, b, mod[:, :D], N, mean, rstd, dres, dx, dmod[:, :D], dmod[:, D:2 * D], dwb,
                                        gate_t=t, gate=mod[:, 2 * D:3 * D], dt=dt, dgate=dmod[:, 2 * D:3 * D]))
print(f"ln_mod_bwd + gate bwd     : {us:7.1f} us  {6 * M * D * 2 / us / 1e6:5.2f} TB/s")

# QK-RMSNorm + RoPE + head split, forward and backward (V in place: only the q and k thirds move)
H, dh = D // 64, 64
qkv = torch.randn(M, 3 * D, device=dev).to(bf)
sq, sk = torch.ones(D, device=dev), torch.ones(D, device=dev)
cs, sn = torch.rand(N, dh // 2, device=dev), torch.rand(N, dh // 2, device=dev)
q, k = torch.empty(B, H, N, dh, device=dev, dtype=bf), torch.empty(B, H, N, dh, device=dev, dtype=bf)
rr = torch.empty(M, 2, device=dev)
us = timeit(lambda: ops.qk_norm_rope_fwd(qkv, sq, sk, cs, sn, q, k, None, rr, B, N, H, dh, dh))
print(f"qk_norm_rope_fwd (V in place): {us:7.1f} us  {4 * M * D * 2 / us / 1e6:5.2f} TB/s")
dq, dk, dqkv, dsc = torch.randn_like(q), torch.randn_like(k), torch.empty_like(qkv), torch.zeros(2, D, device=dev)
us = timeit(lambda: ops.qk_norm_rope_bwd(dq, dk, None, qkv, sq, sk, cs, sn, rr, dqkv, dsc, B, N, H, dh, dh))
print(f"qk_norm_rope_bwd (V in place): {us:7.1f} us  {6 * M * D * 2 / us / 1e6:5.2f} TB/s")
