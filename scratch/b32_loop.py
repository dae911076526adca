"""The reference's default batch (32 molecules, mixed lengths) in a bare loop -- the subject of scratch/b32_timeline.sh.
   python scratch/b32_loop.py [steps] [packed 0/1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import packing_fields, atom_counts
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
model, _ = bench.build_model()
model = model.cuda().train()
if len(sys.argv) > 2 and sys.argv[2] == "1":
    model.strict_reference = False
tuner = FineTuner(model, "classification", total_steps=10000)
NB = int(os.environ.get("B32_BATCH", "32"))
_, batch, label = bench.synth(NB, 128, 256, seed=8765, ragged=True)
host = dict(packing_fields(batch), atom_counts=atom_counts(batch["src_tokens"], 0))
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda(); batch.update(host)
for _ in range(20): tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): tuner.step(batch, label, epoch=0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={NB} layout {model.last_layout}: issue {t_issue / steps * 1e3:.2f} ms/step, wall {t_all / steps * 1e3:.2f} ms/step", flush=True)
if os.environ.get("B32_NO_ADAM"):
    # GPU-bound or host-bound?  drop the optimizer's kernels (0.5 ms of GPU time behind ~0.1 ms of host time) and look at the wall clock
    tuner.optimizer_step = lambda *a, **k: None
    for _ in range(20): tuner.step(batch, label, epoch=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): tuner.step(batch, label, epoch=0)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"  without the optimizer step: issue {t_issue / steps * 1e3:.2f} ms/step, wall {t_all / steps * 1e3:.2f} ms/step", flush=True)
if os.environ.get("B32_SPLIT"):
    # where the host time of a step goes: forward + loss / backward / optimizer, wall clock around each with a device sync in between
    import types
    real = type(tuner).optimizer_step.__get__(tuner)
    tuner.optimizer_step = real
    fb = tuner.forward_backward
    acc = {"fb": 0.0, "opt": 0.0, "fb_sync": 0.0, "opt_sync": 0.0}
    for sync in (False, True):
        for _ in range(steps):
            t = time.perf_counter(); out = fb(batch, label, 0, False)
            if sync: torch.cuda.synchronize()
            acc["fb_sync" if sync else "fb"] += time.perf_counter() - t
            t = time.perf_counter(); real()
            if sync: torch.cuda.synchronize()
            acc["opt_sync" if sync else "opt"] += time.perf_counter() - t
        torch.cuda.synchronize()
    print("  per step (ms): " + ", ".join(f"{k} {v / steps * 1e3:.2f}" for k, v in acc.items()), flush=True)
