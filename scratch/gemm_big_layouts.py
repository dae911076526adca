"""Loop-only rate of gemm_big_kernel for the four operand layouts (is the k-major / tr-read form what holds the weight gradients at ~900 TF/s?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
lib = ops.lib()
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(torch.bfloat16).cuda()
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
lib.mmdti_set_option(b"gemm_big", 2); lib.mmdti_set_option(b"gemm_ring", 0)
for (M, N, K) in [(4096, 4096, 8192), (2048, 2048, 32768)]:
    for tA in (0, 1):
        for tB in (0, 1):
            A = rnd(K, M) if tA else rnd(M, K)
            B = rnd(K, N) if tB else rnd(N, K)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            kw = dict(M=M, N=N, K=K, lda=A.stride(0), ldb=B.stride(0), transA=bool(tA), transB=bool(tB))
            fn = lambda: ops.gemm(A, B, out=out, **kw)
            t = bench(fn)
            lib.mmdti_set_option(b"gemm_dbg", 1); tl = bench(fn); lib.mmdti_set_option(b"gemm_dbg", 0)
            tf = 2.0 * M * N * K / 1e6
            print(f"{M} x {N} x {K} tA={tA} tB={tB}: full {t:7.1f} us ({tf / t:5.0f} TF)   loop only {tl:7.1f} us ({tf / tl:5.0f} TF)", flush=True)
