"""Per-K-tile cost of the 256 x 256 kernel by operand layout: time at K and 4K on 65536 x 512 (two rounds of tiles), the slope is the loop,
the intercept prologue + epilogue.  tA / tB as in ops.gemm; fp16: forward-operand mode; big = 2 forces the 256 x 256 kernel, 0 the 128 x 128 ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
lib = ops.lib()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M, N = 65536, 512
for dt in (torch.bfloat16, torch.float16):
    for tA, tB in ((0, 0), (0, 1)):
        if dt == torch.float16 and tB: continue
        for big in (2, 0):
            lib.mmdti_set_option(b"gemm_big", big)
            row = []
            for K in (512, 2048, 8192):
                A = torch.randn(M, K, device="cuda").to(dt)
                B = (torch.randn(K, N, device="cuda") if tB else torch.randn(N, K, device="cuda")).to(dt)
                kw = dict(M=M, N=N, K=K, lda=A.stride(0), ldb=B.stride(0), transA=False, transB=bool(tB))
                us = t(lambda: ops.gemm(A, B, **kw))
                row.append((K, us, 2.0 * M * N * K / us / 1e6))
            slope = (row[2][1] - row[1][1]) / ((8192 - 2048) / 64)
            print(f"{str(dt)[6:]:9s} tB={tB} big={big}: " + "  ".join(f"K={k}: {u:7.1f} us {tf:6.0f} TF/s" for k, u, tf in row) + f"   loop {slope:.2f} us per K-tile round ({2.0*M*N*64/slope/1e6:.0f} TF/s), intercept {row[1][1]-slope*32:.1f} us")
lib.mmdti_set_option(b"gemm_big", 1)
