import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffulab_amd import ops
dev="cuda"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
M,N,K=256,384,28416
a=(torch.randn(M,K,device=dev)*0.1).to(torch.bfloat16); b=(torch.randn(N,K,device=dev)*0.1).to(torch.bfloat16)
c=torch.empty(M,N,device=dev); scr=torch.empty(1<<23,device=dev)
print("gemm_nt f32 split-K atomics :", round(timeit(lambda: ops.gemm_nt(a,b,c)),1),"us")
print("gemm_nt f32 det (scratch)   :", round(timeit(lambda: ops.gemm_nt(a,b,c,scratch=scr)),1),"us")
print("torch.matmul bf16           :", round(timeit(lambda: torch.matmul(a,b.t())),1),"us")
# wgrad of the stacked adaLN: [28416, 384] += dmod[256, 28416]^T se[256, 384]
dm=(torch.randn(256,28416,device=dev)*0.1).to(torch.bfloat16); se=(torch.randn(256,384,device=dev)*0.1).to(torch.bfloat16)
g=torch.zeros(28416,384,device=dev)
print("gemm_tn mod wgrad atomics   :", round(timeit(lambda: ops.gemm_tn(dm,se,g)),1),"us")
print("gemm_tn mod wgrad det       :", round(timeit(lambda: ops.gemm_tn(dm,se,g,scratch=torch.empty(2*28416*384,device=dev))),1),"us")
