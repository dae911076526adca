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
    sk=ops._splitk_for(N,K,M)
    t_at=bench(lambda: ops.gemm(dy,x,M=N,N=K,K=M,lda=N,ldb=K,transA=True,transB=True,out=dw,ldc=K,atomic=True,splitk=sk))
    res=[f"atomic sk{sk}: {t_at:.0f}"]
    for sp in (sk, 2*sk, 4*sk):
        kc=(M//64//sp)*64
        if kc==0 or kc*sp!=M: 
            # uneven: skip remainder for the timing experiment
            pass
        ws=torch.empty(sp,N,K,device='cuda')
        f=lambda: ops.gemm(dy,x,M=N,N=K,K=kc,lda=N,ldb=K,transA=True,transB=True,out=ws,ldc=K,batch=(1,sp),sA=(0,kc*N),sB=(0,kc*K),sC=(0,N*K),out_dtype=torch.float32)
        t_p=bench(f)
        t_r=bench(lambda: dw.add_(ws.sum(0)))
        res.append(f"partials x{sp} (k={kc}): gemm {t_p:.0f} + torch-reduce {t_r:.0f}")
    print(f"M={M} dW[{N}x{K}]: "+" | ".join(res), flush=True)
