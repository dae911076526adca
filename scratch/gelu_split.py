import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:44s} {s.elapsed_time(e)/iters*1e3:9.1f} us", flush=True)
M=33280
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
x=bf(M,512); w=bf(2048,512); b=torch.randn(2048,device='cuda'); aux=torch.empty(M,2048,device='cuda',dtype=torch.bfloat16)
bench("fc1 plain", lambda: ops.linear_fwd(x,w,b))
bench("fc1 + gelu (no aux)", lambda: ops.linear_fwd(x,w,b,act=ops.ACT_GELU))
bench("fc1 + gelu + aux", lambda: ops.linear_fwd(x,w,b,act=ops.ACT_GELU,aux_out=aux))
bench("fc1 f32 out", lambda: ops.linear_fwd(x,w,b,out_dtype=torch.float32))
dy=bf(M,512); w2=bf(512,2048)
bench("dX N=2048 K=512 plain", lambda: ops.linear_bwd_input(dy,w2))
bench("dX N=2048 K=512 gelu'", lambda: ops.linear_bwd_input(dy,w2,act=ops.ACT_GELU_BWD,aux_in=aux))
cs=torch.zeros(2048,device='cuda')
bench("dX N=2048 K=512 gelu' + colsum", lambda: ops.linear_bwd_input(dy,w2,act=ops.ACT_GELU_BWD,aux_in=aux,colsum=cs))
