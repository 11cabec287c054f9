"""LAB: attention backward + QK-norm backward at the headline shape (B = 256 samples of 256 tokens, 6 heads): the launch pair
dl_attn_bwd_tok + dl_qk_norm_rope_bwd_inplace against the fused dl_attn_bwd_qkn."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops
from diffulab_amd.engine import rope_grid_tables
dev, BF = "cuda", torch.bfloat16
B, H, N, dh = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 6, 256, 64
D, M, scale = H * dh, B * N, dh**-0.5
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
qkv = (torch.randn(M, 3 * D, device=dev)).to(BF)
sq, sk = 1 + 0.1 * torch.randn(D, device=dev), 1 + 0.1 * torch.randn(D, device=dev)
cos, sin = (t.to(dev).contiguous() for t in rope_grid_tables(16, 16, [32, 32], 10_000.0))
q, k = (torch.empty(B, H, N, dh, device=dev, dtype=BF) for _ in range(2))
rrms = torch.empty(M, 2, device=dev)
ops.qk_norm_rope_fwd(qkv, sq, sk, cos, sin, q, k, None, rrms, B, N, H, dh, 64)
out, lse = torch.empty(B, N, D, device=dev, dtype=BF), torch.empty(B, H, N, device=dev)
ops.attn_fwd_qkv(q, k, qkv, out, lse, B, H, N, dh, scale)
do = torch.randn(B, N, D, device=dev).to(BF)
dqkv = torch.empty(M, 3 * D, device=dev, dtype=BF)
ds = torch.zeros(2, D, device=dev)
part = torch.empty(1024 * 2 * D, device=dev)
cpart = torch.empty(B * H * 2 * N, device=dev)
sync = torch.zeros(2 * B + 1, device=dev, dtype=torch.int32)
t_a = timeit(lambda: ops.attn_bwd_tok(q, k, qkv, out, do, lse, dqkv, B, H, N, dh, scale))
t_p = timeit(lambda: (ops.attn_bwd_tok(q, k, qkv, out, do, lse, dqkv, B, H, N, dh, scale),
                      ops.qk_norm_rope_bwd_inplace(qkv, sq, sk, cos, sin, rrms, dqkv, ds, part, B, N, H, dh, 64)))
t_f = timeit(lambda: ops.attn_bwd_qkn(q, k, qkv, out, do, lse, rrms, sq, sk, cos, sin, 64, dqkv, ds, part, cpart, sync, B, H, N, dh, scale))
print(f"{os.environ.get('DIFFULAB_HIP_LIB', 'product').split('_hip_')[-1]:16s} B={B}: attn_bwd_tok {t_a:6.1f} us   pair {t_p:6.1f} us   fused {t_f:6.1f} us   sync clean {int(sync.abs().sum()) == 0}")
