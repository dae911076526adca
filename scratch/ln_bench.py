"""LayerNorm forward / backward standalone at the tower-1 shape [33280, 512]: which of the fused extras cost what."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch
from mmdti_hip import ops
M, D = 33280, 512
x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
dy16 = torch.randn(M, D, device="cuda").to(torch.bfloat16); dy32 = torch.randn(M, D, device="cuda"); dres = torch.randn(M, D, device="cuda")
dg, db, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
_, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
# a working set larger than the 256 MB cache between repetitions
spoil = torch.empty(128 << 20, device="cuda")
def once(fn, reps=10):
    ts = []
    for _ in range(reps):
        spoil.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
cases = {
  "fwd (bf16 out)": (lambda: ops.layernorm_fwd(x, g, b, 1e-5), M * D * 6),
  "fwd (fp32 + bf16 out)": (lambda: ops.layernorm_fwd(x, g, b, 1e-5, want_f32=True), M * D * 10),
  "bwd dy bf16": (lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dg, db), M * D * 10),
  "bwd dy bf16 + dres": (lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dg, db, dres=dres), M * D * 14),
  "bwd dy bf16 + dres + bf16 copy": (lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dg, db, dres=dres, bf16_copy=(0.0, 1, None)), M * D * 16),
  "bwd dy bf16 + dres + bf16 copy p=0.1": (lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dg, db, dres=dres, bf16_copy=(0.1, 1, None)), M * D * 16),
  "bwd dy bf16 + dres + bf16 copy p=0.1 + colsum": (lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dg, db, dres=dres, bf16_copy=(0.1, 1, cs)), M * D * 16),
  "bwd dy fp32 + dres": (lambda: ops.layernorm_bwd(dy32, x, g, mean, rstd, dg, db, dres=dres), M * D * 16),
  "torch copy fp32 (r+w)": (lambda: dres.copy_(dy32), M * D * 8),
}
for name, (fn, byts) in cases.items():
    for _ in range(3): fn()
    t = once(fn)
    print(f"{name:48s} {t:6.1f} us  {byts / t / 1e6:5.2f} TB/s", flush=True)
