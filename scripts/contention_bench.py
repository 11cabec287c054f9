"""how much the main chain's kernels slow down next to a side-stream weight-gradient GEMM (the training step's overlap):
each kernel is timed alone and again while gemm_tn (MLP-up shape, capped at 192 workgroups like the engine does) runs on a second
stream.   python scripts/contention_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops

dev, bf = "cuda", torch.bfloat16
B, N, D = 256, 256, 384
M = B * N
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf)  # noqa: E731
x, t, dout, dres = rnd(M, D), rnd(M, D), rnd(M, D), rnd(M, D)
mod = rnd(B, 6 * D)
w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
out, dx, dt = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
ops.ln_modulate_fwd(x, w, b, mod[:, :D], mod[:, D:2 * D], N, 1e-5, out, mean, rstd)
dmod, dwb = torch.zeros(B, 6 * D, device=dev), torch.zeros(B, 2, D, device=dev)
u, dh, du = rnd(M, 8 * D), rnd(M, 4 * D), torch.empty(M, 8 * D, device=dev, dtype=bf)
xm2, gW = rnd(M, D), torch.zeros(8 * D, D, device=dev)
wt = rnd(D, 8 * D)
dxm = torch.empty(M, D, device=dev, dtype=bf)

kernels = {
    "ln_mod_bwd+gate": lambda: ops.ln_modulate_bwd(dout, x, w, b, mod[:, :D], N, mean, rstd, dres, dx, dmod[:, :D], dmod[:, D:2 * D], dwb,
                                                   gate_t=t, gate=mod[:, 2 * D:3 * D], dt=dt, dgate=dmod[:, 2 * D:3 * D]),
    "ln_mod_fwd+resid": lambda: ops.ln_modulate_fwd(x, w, b, mod[:, :D], mod[:, D:2 * D], N, 1e-5, out, mean, rstd, t=t,
                                                    gate=mod[:, 2 * D:3 * D], x_out=dx),
    "swiglu_bwd": lambda: ops.swiglu_bwd(dh, u, du),
    "gemm_nt d_xm2": lambda: ops.gemm_nt(du, wt, dxm),
}
side = torch.cuda.Stream()


def timeit(fn, busy: bool, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if busy:
        with torch.cuda.stream(side):
            for _ in range(4 * iters):
                ops.gemm_tn(du, xm2, gW, max_wgs=192)
        torch.cuda._sleep(200000)  # let the side stream get going
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, fn in kernels.items():
    print(f"{name:18s}: alone {timeit(fn, False):7.1f} us   beside gemm_tn {timeit(fn, True):7.1f} us", flush=True)
