"""Standalone timing of the fused BERT attention kernels at the bench shape (B=256, 8 heads x 64, 256 x 256, dropout 0.1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
B, heads, L, hd = 256, 8, 256, 64
D = heads * hd
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = (torch.randn(B * L, D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(4))
add = torch.zeros(B, L, device="cuda")
scale = hd ** -0.5
def t(fn, n=iters):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
ctx, stats = ops.attn_fwd(q, k, v, add, B, heads, L, L, scale, p, 1, 1)
tf = t(lambda: ops.attn_fwd(q, k, v, add, B, heads, L, L, scale, p, 1, 1))
tb = t(lambda: ops.attn_bwd(q, k, v, add, do, stats, B, heads, L, L, scale, p, 1, 1))
fl = 4.0 * L * L * hd * B * heads
print(f"p={p}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s)   bwd (dQ + dK/dV kernels) {tb:.1f} us ({3.5 * fl / tb / 1e6:.0f} TF/s)")
