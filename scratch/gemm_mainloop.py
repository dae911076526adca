import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
def bench(name, fn, flops, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s", flush=True)
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for (M,N,K) in [(32768,512,512),(32768,512,2048),(32768,512,8192),(32768,2048,512),(32768,2048,2048),(32768,2048,8192),(8192,8192,8192)]:
    x,w=bf(M,K),bf(N,K)
    bench(f"NT M={M} N={N} K={K}", lambda: ops.linear_fwd(x,w), 2*M*N*K)
    bench(f"  torch (hipBLASLt)", lambda: torch.nn.functional.linear(x,w), 2*M*N*K)
