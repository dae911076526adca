"""Fused Linear + residual + LayerNorm kernel (mmdti_gemm_ln_bf16) against the two kernels it replaces, on the step's shapes.
   python scratch/gemm_ln_bench.py   ->  gpurun_out/gemm_ln_ab.json"""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from mmdti_hip import ops

def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

rows = []
for M, K, f32 in ((33280, 512, False), (33280, 2048, False), (65536, 512, True), (65536, 2048, True), (12713, 512, False), (12713, 2048, False), (9823, 2048, True), (6400, 512, False), (4800, 512, False), (3200, 512, False), (1600, 512, False), (1600, 2048, False)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(512, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(512, device="cuda"); res = torch.randn(M, 512, device="cuda"); g = torch.ones(512, device="cuda"); be = torch.zeros(512, device="cuda")
    def fused(): ops.linear_ln_fwd(x, w, b, g, be, 1e-5, residual=res, drop_p=0.1, seed=1, site=1, want_f32=f32, want_bf16=True)
    def gemm(): return ops.linear_fwd(x, w, b, residual=res, out_dtype=torch.float32, drop_p=0.1, seed=1, site=1)
    y = gemm()
    def ln(): ops.layernorm_fwd(y, g, be, 1e-5, want_f32=f32, want_bf16=True)
    def two(): ops.layernorm_fwd(gemm(), g, be, 1e-5, want_f32=f32, want_bf16=True)
    r = dict(M=M, K=K, f32=f32, fused_us=round(bench(fused), 1), gemm_us=round(bench(gemm), 1), ln_us=round(bench(ln), 1), two_us=round(bench(two), 1))
    os.environ["X"] = "1"
    rows.append(r); print(r)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "gemm_ln_ab.json"), "w"), indent=1)
