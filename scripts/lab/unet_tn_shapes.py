"""LAB: the (M, N, R) of every dl_gemm_tn_ex call of one MNIST-DDPM UNet training step, with each distinct shape timed alone."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import Diffuser, ops
from diffulab_amd.networks.denoisers import UNetModel
from diffulab_amd.training.optim import FusedAdamW

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
orig = ops.gemm_tn
def spy(a, b, out, *, M=None, N=None, max_wgs=0, scratch=None):
    seen[(a.shape[1] if M is None else M, b.shape[1] if N is None else N, (a.shape[0] + 63) // 64 * 64, a.stride(0), b.stride(0))] += 1
    return orig(a, b, out, M=M, N=N, max_wgs=max_wgs, scratch=scratch)
ops.gemm_tn = spy
import diffulab_amd.unet_engine as ue
ue.ops.gemm_tn = spy
step(); torch.cuda.synchronize()
ops.gemm_tn = orig; ue.ops.gemm_tn = orig
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = tot0 = 0.0
for (M, N, R, lda, ldb), cnt in sorted(seen.items(), key=lambda kv: (-kv[0][2], kv[0])):
    a = torch.randn(R, lda, device=dev).to(torch.bfloat16); b = torch.randn(R, ldb, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev)
    ops.lib().cdll.dl_lab_set_tn_split_model(0)
    t0 = timeit(lambda: orig(a, b, out, M=M, N=N))
    ops.lib().cdll.dl_lab_set_tn_split_model(1)
    t = timeit(lambda: orig(a, b, out, M=M, N=N))
    tot += t * cnt; tot0 += t0 * cnt
    print(f"M={M:5d} N={N:5d} R={R:6d} lda={lda:5d} ldb={ldb:5d} x{cnt:2d}: workgroup-count rule {t0:6.1f} us   split model {t:6.1f} us  {2.0 * M * N * R / t / 1e6:6.1f} TF/s   out {M * N * 4 / 1e6:5.2f} MB")
print(f"sum over the step: {tot0:.0f} -> {tot:.0f} us")
