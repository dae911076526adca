import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
import os
B,N,H=256,int(os.environ.get('PA_N','130')),64; D=H*8; ld=ops.pair_ld(N); scale=8**-0.5
p=float(sys.argv[1]) if len(sys.argv)>1 else 0.1
iters=int(sys.argv[2]) if len(sys.argv)>2 else 10
qkv=(torch.randn(B*N,3*D,device='cuda')).to(torch.bfloat16)
tiled=len(sys.argv)>3 and sys.argv[3]=='tiled'
bias=torch.randn(B,H,N,ld,device='cuda')
if tiled: bias=ops.pair_tile(bias,N)
if len(sys.argv)>4 and sys.argv[4]=='f16': bias=bias.half()
gdt=torch.bfloat16 if (len(sys.argv)>5 and sys.argv[5]=='g16') else torch.float32
do=torch.randn(B*N,D,device='cuda').to(torch.bfloat16)
def t(fn,n=iters):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
s_out,o=ops.pair_attn_fwd(qkv,bias,None,B,N,H,ld,scale,p,1,1)
g=torch.zeros(bias.shape,device='cuda',dtype=gdt)
print("fwd ms", t(lambda: ops.pair_attn_fwd(qkv,bias,None,B,N,H,ld,scale,p,1,1)))
print("bwd ms (g_in)", t(lambda: ops.pair_attn_bwd(qkv,s_out,do,g,B,N,H,ld,scale,False,p,1,1)))
print("bwd ms (g zero)", t(lambda: ops.pair_attn_bwd(qkv,s_out,do,g,B,N,H,ld,scale,True,p,1,1)))
