"""LAB: the (M, N, K) of every dl_gemm_nt call of one MNIST-DDPM UNet training step, each distinct shape timed alone (L2-hot and
with a 512 MB buffer swept between launches: operands cold, what a launch inside the step sees)."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import Diffuser, ops
from diffulab_amd.networks.denoisers import UNetModel
from diffulab_amd.training.optim import FusedAdamW
import diffulab_amd.unet_engine as ue

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
torch.manual_seed(0)
m = UNetModel(image_size=[32, 32], in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2, attention_resolutions=[4, 8, 16],
              num_heads=2, resblock_updown=True, n_classes=10, use_scale_shift_norm=True, classifier_free=False).to(dev)
gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
opt = FusedAdamW(m.parameters(), lr=1e-4)
x0 = torch.randn(B, 1, 32, 32, device=dev); y = torch.randint(0, 10, (B,), device=dev)
def step():
    opt.zero_grad()
    gd.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=gd.draw_timesteps(B))["loss"].backward()
    opt.step()
for _ in range(2): step()
seen = collections.Counter()
orig = ops.gemm_nt
def spy(a, b, out, *, bias=None, act=0, pre_out=None, resid=None, gate=None, rows_per_gate=1, M=None, N=None, K=None, scratch=None):
    seen[(a.shape[0] if M is None else M, b.shape[0] if N is None else N, a.shape[1] if K is None else K, str(out.dtype)[6:], bias is not None, resid is not None)] += 1
    return orig(a, b, out, bias=bias, act=act, pre_out=pre_out, resid=resid, gate=gate, rows_per_gate=rows_per_gate, M=M, N=N, K=K, scratch=scratch)
ops.gemm_nt = spy
step(); torch.cuda.synchronize()
ops.gemm_nt = orig
flush = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
def timeit(fn, n=20, cold=False):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if cold: flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1e3 / n
th = tc = 0.0
for (M, N, K, dt, hb, hr), cnt in sorted(seen.items(), key=lambda kv: (-kv[0][0], kv[0])):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if dt == "float32" else torch.bfloat16)
    bias = torch.randn(N, device=dev) if hb else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if hr else None
    t1 = timeit(lambda: orig(a, w, out, bias=bias, resid=res))
    t2 = timeit(lambda: orig(a, w, out, bias=bias, resid=res), cold=True)
    th += t1 * cnt; tc += t2 * cnt
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M:6d} N={N:5d} K={K:5d} {dt:8s} bias={int(hb)} resid={int(hr)} x{cnt:2d} tiles128 {tiles:5d}: hot {t1:6.1f} us  cold {t2:6.1f} us  {2.0 * M * N * K / t2 / 1e6:6.1f} TF/s")
print(f"sum over the step: hot {th:.0f} us, cold {tc:.0f} us")
