import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e)/iters*1e3:9.1f} us", flush=True)
B,N,K,Fh,H,E=256,130,128,128,64,961
ld=ops.pair_ld(N)
g=torch.Generator().manual_seed(1)
dist=(torch.rand(B,N,N,generator=g)*8).cuda(); et=torch.randint(0,E,(B,N,N),generator=g).cuda()
mul=(1+0.1*torch.randn(E,generator=g)).cuda(); bias=(0.1*torch.randn(E,generator=g)).cuda()
means=(torch.rand(K,generator=g)*3).cuda(); stds=(torch.rand(K,generator=g)*3-1.5).cuda()
w1=(torch.randn(Fh,K,generator=g)*0.2).cuda().bfloat16(); b1=(torch.randn(Fh,generator=g)*0.1).cuda()
w2=(torch.randn(H,Fh,generator=g)*0.2).cuda().bfloat16(); b2=(torch.randn(H,generator=g)*0.1).cuda()
for sg in (False, True):
    bench(f"fused fwd tiled save_grad={sg}", lambda: ops.gbf_bias_fwd(dist,et,mul,bias,means,stds,w1,b1,w2,b2,ld,save=True,tiled=True,save_grad=sg))
    out,(feat,u,h)=ops.gbf_bias_fwd(dist,et,mul,bias,means,stds,w1,b1,w2,b2,ld,save=True,tiled=True,save_grad=sg)
    G=torch.randn_like(out); G[torch.isinf(out)]=0
    gr=[torch.zeros_like(t) for t in (mul,bias,means,stds)]
    bench(f"fused bwd u_is_grad={sg}", lambda: ops.gbf_bias_bwd(G,dist,et,mul,bias,means,stds,w1,w2,u,ld,*gr,u_is_grad=sg))
