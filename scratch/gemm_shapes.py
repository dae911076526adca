"""Per-shape GEMM time inside the bench step (every stream overlap off: each launch alone on the chip), with the
roofline bound of each shape: max(2MNK / 2.5 PF, bytes / 6.3 TB/s)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip import ops
from mmdti_hip import functional as Fn
from mmdti_hip.trainer import FineTuner
model, _ = bench.build_model()
model = model.cuda().train()
model.overlap_towers = False; model.infonce_on_side_stream = False; model.cross_modal_module.two_streams = False; Fn.DEFER_WGRAD_LAYERS = 0
tuner = FineTuner(model, "classification", total_steps=1000)
GS_B, GS_RAG = int(os.environ.get("GS_BATCH", "256")), os.environ.get("GS_RAGGED") == "1"   # GS_BATCH=32 GS_RAGGED=1: the small-batch step
_, batch, label = bench.synth(GS_B, 128, 256, seed=1234, ragged=GS_RAG)
host = {}
if GS_RAG:
    from mmdti_hip.collate import packing_fields, atom_counts
    host = dict(packing_fields(batch), atom_counts=atom_counts(batch["src_tokens"], 0))
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda()
batch.update(host)
for _ in range(3): tuner.step(batch, label)
torch.cuda.synchronize()
ops.kernel_timer.enable(("gemm",))
for _ in range(3): tuner.step(batch, label)
rows = ops.kernel_timer.by_tag("gemm")
ops.kernel_timer.disable()
tot = 0.0
out = []
for tag, d in rows.items():
    if tag[0] == "grouped_dw":
        flops = d["work"] / d["n"]
        us = d["mean_ms"] * 1e3
        byts = sum(2.0 * tag[2] * (a + b) + 8.0 * a * b for a, b in tag[1])
        t_mfma, t_hbm = flops / 2.5e15 * 1e6, byts / 6.3e12 * 1e6
        print("grouped dW", tag[1], "rows", tag[2], f"n/step {d['n'] / 3:.1f}  {us:.1f} us  {flops / us / 1e6:.0f} TF/s  bound {max(t_mfma, t_hbm):.1f} us  ms/step {d['total_ms'] / 3:.3f}")
        out.append(dict(M=0, N=0, K=tag[2], tA=1, tB=1, batch=1, splitk=0, act=0, n_per_step=d["n"] / 3, us=us, tflops=flops / us / 1e6, bound_us=max(t_mfma, t_hbm),
                        bound="mfma", ms_per_step=d["total_ms"] / 3, frac_of_bound=max(t_mfma, t_hbm) / us, grouped=str(tag[1])))
        continue
    if tag[7] == "ln":
        M, N, K = tag[:3]; tA = tB = 0; nb = 1; sk = 1; act = 9; obf = False; aux = True; res = tag[10]
    else:
        M, N, K, tA, tB, nb, sk, act, obf, aux, res = tag
    flops = 2.0 * M * N * K * nb
    byts = 2.0 * nb * (M * K + N * K) + (2 if obf else 4) * nb * M * N * (2 if (res or sk > 1) else 1) + (2 * M * N if aux else 0)
    t_mfma, t_hbm = flops / 2.5e15 * 1e6, byts / 6.3e12 * 1e6
    us = d["mean_ms"] * 1e3
    out.append(dict(M=M, N=N, K=K, tA=tA, tB=tB, batch=nb, splitk=sk, act=act, n_per_step=d["n"] / 3, us=us, tflops=flops / us / 1e6, bound_us=max(t_mfma, t_hbm),
                    bound="mfma" if t_mfma > t_hbm else "hbm", ms_per_step=d["total_ms"] / 3, frac_of_bound=max(t_mfma, t_hbm) / us))
out.sort(key=lambda r: -r["ms_per_step"])
print(f"{'M':>6} {'N':>5} {'K':>6} tA tB  sk act  n/step     us   TF/s  bound_us bound  frac  ms/step")
for r in out:
    print(f"{r['M']:6d} {r['N']:5d} {r['K']:6d}  {r['tA']}  {r['tB']} {r['splitk']:3d} {r['act']:3d} {r['n_per_step']:7.1f} {r['us']:7.1f} {r['tflops']:6.0f} {r['bound_us']:9.1f} {r['bound']:>5} {r['frac_of_bound']:5.2f} {r['ms_per_step']:8.3f}")
print("total GEMM ms/step", sum(r["ms_per_step"] for r in out), " at-bound ms/step", sum(r["bound_us"] * r["n_per_step"] for r in out) / 1e3)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.json"), "w"), indent=1)

import csv
with open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.csv"), "w") as f:
    f.write("# per-shape GEMM launches of the headline step (scratch/gemm_shapes.py: HIP events around every launch, every stream overlap off); bound_us = "
            "max(flop / 2.5 PF, algorithmic bytes / 6.3 TB/s); frac_of_bound = bound_us / us\n")
    w = csv.DictWriter(f, fieldnames=["M", "N", "K", "tA", "tB", "batch", "splitk", "act", "n_per_step", "us", "tflops", "bound_us", "bound", "frac_of_bound", "ms_per_step", "grouped"])
    w.writeheader()
    for r in out:
        w.writerow({k: (f"{v:.3f}" if isinstance(v, float) else v) for k, v in dict(r, grouped=r.get("grouped", "")).items()})
