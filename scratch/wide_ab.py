import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e3
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
M=65536
out=[]
for (N,K,kind) in [(512,2048,'fwd'),(512,1536,'dX'),(512,2048,'dX'),(512,1024,'fwd'),(1536,1024,'fwd')]:
    if kind=='fwd':
        x,w=bf(M,K),bf(N,K); t=bench(lambda: ops.linear_fwd(x,w))
    else:
        dy,w=bf(M,K),bf(K,N); t=bench(lambda: ops.linear_bwd_input(dy,w))
    out.append(f"{kind} N={N} K={K}: {t:.1f}")
print(" | ".join(out))
