import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, flops, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    print(f"{name:40s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s", flush=True)
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for K in (512,2048):
  for N in (512,):
    for M in (128*256//4, 128*512//4, 128*1024//4, 128*1040//4, 128*1536//4, 128*2048//4):
        x,w,b=bf(M,K),bf(N,K),torch.randn(N,device='cuda')
        bench(f"fwd M={M} ({M//128*N//128} tiles) N={N} K={K}", lambda: ops.linear_fwd(x,w,b), 2*M*N*K)
