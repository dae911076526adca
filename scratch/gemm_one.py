import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
M,N,K=33280,512,2048
x=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=torch.randn(N,K,device='cuda').to(torch.bfloat16); b=torch.randn(N,device='cuda')
for _ in range(5): ops.linear_fwd(x,w,b)
torch.cuda.synchronize()
