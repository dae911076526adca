"""grouped weight gradients only: x bf16 vs x fp16 (in-kernel conversion), interleaved"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from mmdti_hip import ops
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(*s): return torch.randn(*s, device="cuda", generator=g)
for rows, shapes in ((33280, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]), (65536, [(1536, 512), (512, 2048), (2048, 512), (512, 512)])):
    base = [(rnd(rows, no).bfloat16(), rnd(rows, ni), torch.zeros(no, ni, device="cuda"), torch.zeros(no, device="cuda")) for no, ni in shapes]
    res = {}
    for rep in range(3):
        for tag, cv in (("bf16", lambda x: x.bfloat16()), ("fp16", lambda x: x.half())):
            items = [(dy, cv(x), dw, db, None) for dy, x, dw, db in base]
            res.setdefault(tag, []).append(timeit(lambda: ops.linear_bwd_weight_grouped(items)))
    print(f"{os.environ.get('MMDTI_HIP_LIB', 'in-tree')[-12:]} rows {rows:6d}: bf16 {min(res['bf16']):7.1f} us   fp16 {min(res['fp16']):7.1f} us   x {min(res['fp16']) / min(res['bf16']):.3f}")
