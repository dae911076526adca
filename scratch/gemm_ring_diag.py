import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
lib = ops.lib()
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(torch.bfloat16).cuda()
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K, tB) in [(65536, 512, 2048, 0), (65536, 512, 2048, 1), (65536, 512, 512, 0), (65536, 2048, 512, 0), (32768, 512, 2048, 0), (16384, 512, 2048, 0), (8192, 512, 2048, 0),
                      (65536, 512, 8192, 0), (33280, 512, 2048, 0), (8192, 8192, 8192, 0)]:
    A = rnd(M, K); B = rnd(K, N) if tB else rnd(N, K)
    kw = dict(M=M, N=N, K=K, lda=A.stride(0), ldb=B.stride(0), transB=bool(tB))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm(A, B, out=out, **kw)
    lib.mmdti_set_option(b"gemm_ring", 0); t_old = bench(fn)
    lib.mmdti_set_option(b"gemm_ring", 2); t_ring = bench(fn)
    lib.mmdti_set_option(b"gemm_dbg", 1); t_loop = bench(fn); lib.mmdti_set_option(b"gemm_dbg", 0)
    tf = 2.0 * M * N * K / 1e6
    tiles = -(-M // 256) * -(-N // 128)
    print(f"{M:6d} {N:5d} {K:5d} tB={tB} tiles {tiles:5d}  old {t_old:7.1f} ({tf/t_old:5.0f} TF)  ring {t_ring:7.1f} ({tf/t_ring:5.0f} TF)  ring loop only {t_loop:7.1f} ({tf/t_loop:5.0f} TF)", flush=True)
