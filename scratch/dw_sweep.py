import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e3
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for M in (33280, 65536):
  for (N,K) in [(512,512),(1536,512),(2048,512),(512,2048)]:
    dy,x=bf(M,N),bf(M,K); dw=torch.zeros(N,K,device='cuda')
    tiles=(N//128)*(K//128)
    res=[]
    for sk in (4,6,8,12,16,24,32,48,64,96):
        if tiles*sk>4096: continue
        t=bench(lambda: ops.gemm(dy,x,M=N,N=K,K=M,lda=N,ldb=K,transA=True,transB=True,out=dw,ldc=K,atomic=True,splitk=sk))
        res.append(f"sk{sk}({tiles*sk}):{t:.0f}")
    print(f"M={M} dW[{N}x{K}] tiles={tiles}: "+"  ".join(res), flush=True)
