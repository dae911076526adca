"""What do fp16 forward operands cost per kernel?  Interleaved timings (HIP events, alone on the chip) of the step's GEMM launches with
bf16 and with fp16 operands: forward Linears (fp16 A and B), grouped weight gradients (fp16 x converted inside the kernel)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from mmdti_hip import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

g = torch.Generator(device="cuda").manual_seed(1)
def rnd(*s, scale=1.0): return torch.randn(*s, device="cuda", generator=g) * scale

print("== grouped weight gradients (x bf16 vs x fp16 converted in the kernel), us")
for rows, shapes in ((33280, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]), (65536, [(1536, 512), (512, 2048), (2048, 512), (512, 512)]),
                     (12713, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]), (1695, [(512, 2048), (2048, 512), (512, 512), (1536, 512)])):
    base = [(rnd(rows, no).bfloat16(), rnd(rows, ni), torch.zeros(no, ni, device="cuda"), torch.zeros(no, device="cuda")) for no, ni in shapes]
    res = {}
    for rep in range(3):
        for tag, cv in (("bf16", lambda x: x.bfloat16()), ("fp16", lambda x: x.half())):
            items = [(dy, cv(x), dw, db, None) for dy, x, dw, db in base]
            res.setdefault(tag, []).append(timeit(lambda: ops.linear_bwd_weight_grouped(items)))
    flop = sum(2.0 * rows * no * ni for no, ni in shapes)
    print(f"rows {rows:6d}: bf16 {min(res['bf16']):7.1f} us ({flop / min(res['bf16']) / 1e6:6.0f} TF/s)   fp16 {min(res['fp16']):7.1f} us ({flop / min(res['fp16']) / 1e6:6.0f} TF/s)   "
          f"x {min(res['fp16']) / min(res['bf16']):.3f}")

print("== forward Linears (bf16 vs fp16 operands and outputs), us")
for M, N, K, act in ((33280, 1536, 512, 0), (33280, 2048, 512, 1), (33280, 512, 2048, 0), (33280, 512, 512, 0), (65536, 1536, 512, 0), (65536, 2048, 512, 1), (65536, 512, 2048, 0),
                     (65536, 512, 512, 0)):
    x, w, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N)
    res = {}
    for rep in range(3):
        for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            xx, ww = x.to(dt), w.to(dt)
            u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            if act: fn = lambda: ops.linear_fwd(xx, ww, b, act=ops.ACT_GELU_FWD, aux_out=u)
            elif N == 512:
                res32 = rnd(M, N)
                fn = lambda: ops.linear_fwd(xx, ww, b, residual=res32, out_dtype=torch.float32, drop_p=0.1, seed=1, site=1)
            else: fn = lambda: ops.linear_fwd(xx, ww, b)
            res.setdefault(tag, []).append(timeit(fn))
    flop = 2.0 * M * N * K
    print(f"{M:6d} x {N:4d} x {K:4d} act {act}: bf16 {min(res['bf16']):7.1f} us ({flop / min(res['bf16']) / 1e6:6.0f} TF/s)   fp16 {min(res['fp16']):7.1f} us   x {min(res['fp16']) / min(res['bf16']):.3f}")
print("== fused Linear + LayerNorm, us")
for M, K in ((33280, 512), (65536, 512)):
    x, w, b, r = rnd(M, K), rnd(512, K, scale=0.05), rnd(512), rnd(M, 512)
    gam, bet = torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
    res = {}
    for rep in range(3):
        for tag, f16 in (("bf16", False), ("fp16", True)):
            ops.set_forward_fp16(f16)
            xx, ww = x.to(ops.act16()), w.to(ops.act16())
            res.setdefault(tag, []).append(timeit(lambda: ops.linear_ln_fwd(xx, ww, b, gam, bet, 1e-5, residual=r, drop_p=0.1, seed=1, site=1)))
    print(f"{M:6d} x 512 x {K:4d}: bf16 {min(res['bf16']):7.1f} us   fp16 {min(res['fp16']):7.1f} us   x {min(res['fp16']) / min(res['bf16']):.3f}")
