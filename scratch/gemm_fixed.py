import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:50s} {s.elapsed_time(e)/iters*1e3:8.1f} us", flush=True)
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
N=512
for M in (8192*3, 32768, 33280):
  for K in (64,128,256,512,1024):
    x,w,b=bf(M,K),bf(N,K),torch.randn(N,device='cuda')
    out=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
    bench(f"M={M} tiles={M//128*4} K={K} bf16 out", lambda: ops.gemm(x,w,M=M,N=N,K=K,lda=K,ldb=K,bias=b,out=out,ldc=N))
