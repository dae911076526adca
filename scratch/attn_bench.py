import sys, math, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:30s} {s.elapsed_time(e)/iters*1e3:8.1f} us", flush=True)
B,heads,L,hd=256,8,256,64; D=heads*hd
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
q,k,v,do=bf(B*L,D),bf(B*L,D),bf(B*L,D),bf(B*L,D)
add=torch.zeros(B,L,device='cuda')
sc=1/math.sqrt(hd)
ctx,st=ops.attn_fwd(q,k,v,add,B,heads,L,L,sc,0.1,1,1)
bench("attn fwd", lambda: ops.attn_fwd(q,k,v,add,B,heads,L,L,sc,0.1,1,1))
bench("attn bwd (q+kv)", lambda: ops.attn_bwd(q,k,v,add,do,st,B,heads,L,L,sc,0.1,1,1))
