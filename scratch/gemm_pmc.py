import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
M=33280
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for (N,K) in [(512,512),(2048,512),(512,2048)]:
    x,w,b=bf(M,K),bf(N,K),torch.randn(N,device='cuda')
    dy=bf(M,N); dw=torch.zeros(N,K,device='cuda')
    for _ in range(3):
        ops.linear_fwd(x,w,b); ops.linear_bwd_input(dy,w); ops.linear_bwd_weight(dy,x,dw)
torch.cuda.synchronize()
