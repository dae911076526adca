import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
M,N,K=32768,2048,2048
x,w=bf(M,K),bf(N,K)
for _ in range(3): ops.linear_fwd(x,w)
torch.cuda.synchronize()
