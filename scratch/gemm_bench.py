import sys, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s")
M=int(sys.argv[1]) if len(sys.argv)>1 else 33280
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for (N,K) in [(1536,512),(512,512),(2048,512),(512,2048)]:
    x,w,b=bf(M,K),bf(N,K),torch.randn(N,device='cuda')
    bench(f"fwd NT  M={M} N={N} K={K} bf16out", lambda: ops.linear_fwd(x,w,b), 2*M*N*K)
    dy=bf(M,N)
    bench(f"dX  NN  M={M} N={K} K={N}", lambda: ops.linear_bwd_input(dy,w), 2*M*N*K)
    dw=torch.zeros(N,K,device='cuda')
    bench(f"dW  TN  M={N} N={K} K={M}", lambda: ops.linear_bwd_weight(dy,x,dw), 2*M*N*K)
res=torch.randn(M,512,device='cuda'); x=bf(M,2048); w=bf(512,2048); b=torch.randn(512,device='cuda')
bench("fc2 + residual + dropout f32out", lambda: ops.linear_fwd(x,w,b,residual=res,out_dtype=torch.float32,drop_p=0.1,seed=1,site=1), 2*M*512*2048)
x=bf(M,512); w=bf(2048,512); b=torch.randn(2048,device='cuda'); aux=torch.empty(M,2048,device='cuda',dtype=torch.bfloat16)
bench("fc1 + gelu + aux", lambda: ops.linear_fwd(x,w,b,act=ops.ACT_GELU,aux_out=aux), 2*M*512*2048)
# gbf proj shapes
P=256*130*130
x=bf(P,128); w=bf(128,128); b=torch.randn(128,device='cuda'); aux=torch.empty(P,128,device='cuda',dtype=torch.bfloat16)
bench("gbf linear1 P x128x128 gelu", lambda: ops.linear_fwd(x,w,b,act=ops.ACT_GELU,aux_out=aux), 2*P*128*128, iters=5)
w2=bf(64,128); b2=torch.randn(64,device='cuda')
bench("gbf linear2 P x64x128 f32out", lambda: ops.linear_fwd(x,w2,b2,out_dtype=torch.float32), 2*P*64*128, iters=5)
