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
    print(f"{name:44s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s", flush=True)
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
M=33280
for (N,K) in [(512,512),(512,2048),(1536,512),(2048,512),(512,1536)]:
    x,w,b=bf(M,K),bf(N,K),torch.randn(N,device='cuda')
    ref=ops.linear_fwd(x,w,b)
    bench(f"fwd M={M} N={N} K={K}", lambda: ops.linear_fwd(x,w,b), 2*M*N*K)
    wt=bf(N,K)  # dX: dy [M,N] . w [N,K] -> [M,K]
    dy=bf(M,N)
    bench(f"dX  M={M} N={K} K={N}", lambda: ops.linear_bwd_input(dy,wt), 2*M*N*K)
