import torch
def bench(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s", flush=True)
bf=lambda *s: torch.randn(*s,device='cuda').to(torch.bfloat16)
for M in (33280, 65536):
  for (N,K) in [(1536,512),(512,512),(2048,512),(512,2048)]:
    x,w=bf(M,K),bf(N,K); dy=bf(M,N)
    bench(f"lib fwd NT M={M} N={N} K={K}", lambda: torch.nn.functional.linear(x,w), 2*M*N*K)
    bench(f"lib dX  NN M={M} N={K} K={N}", lambda: dy@w, 2*M*N*K)
    bench(f"lib dW  TN M={N} N={K} K={M}", lambda: dy.t()@x, 2*M*N*K)
