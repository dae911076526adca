"""What the fused epilogues cost on the step's GEMM shapes: no epilogue / bias / GELU / GELU + saved gelu' / fp32 + residual.
Variants are timed INTERLEAVED (5 rounds, median) -- timed back to back once each, run-order effects of +-10 % look like epilogue costs."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(torch.bfloat16).cuda()
def once(fn, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M in (33280, 65536):
    for (N, K) in ((2048, 512), (1536, 512), (512, 512), (512, 2048)):
        A, B = rnd(M, K), rnd(N, K)
        bias = torch.randn(N).cuda()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(out)
        out32 = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda")
        kw = dict(M=M, N=N, K=K, lda=K, ldb=K)
        fns = {"plain": lambda: ops.gemm(A, B, out=out, **kw),
               "bias": lambda: ops.gemm(A, B, out=out, bias=bias, **kw),
               "gelu": lambda: ops.gemm(A, B, out=out, bias=bias, act=ops.ACT_GELU, **kw),
               "gelu_g+aux": lambda: ops.gemm(A, B, out=out, bias=bias, act=ops.ACT_GELU_G, aux_out=aux, **kw),
               "f32+res": lambda: ops.gemm(A, B, out=out32, bias=bias, residual=res, out_dtype=torch.float32, **kw)}
        for fn in fns.values():
            for _ in range(3): fn()
        t = {k: [] for k in fns}
        for _ in range(5):
            for k, fn in fns.items(): t[k].append(once(fn))
        tf = 2.0 * M * N * K / 1e6
        print(f"{M:6d} {N:5d} {K:5d}  " + "  ".join(f"{k} {statistics.median(v):6.1f} us ({tf / statistics.median(v):4.0f} TF)" for k, v in t.items()), flush=True)
