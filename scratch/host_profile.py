"""Host-side cost of one eager step at a small batch (the reference's default 32 molecules, mixed lengths): cProfile over
50 steps, sorted by own time -- where the Python / ctypes / allocator microseconds of ~480 launches per step go.
   python scratch/host_profile.py [batch]   ->  gpurun_out/host_profile.txt"""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import packing_fields, atom_counts
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, _ = bench.build_model()
model = model.cuda().train()
tuner = FineTuner(model, "classification", total_steps=10000)
_, batch, label = bench.synth(B, 128, 256, seed=1234, ragged=True)
host = dict(packing_fields(batch), atom_counts=atom_counts(batch["src_tokens"], 0))
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda(); batch.update(host)
for _ in range(20): tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): tuner.step(batch, label, epoch=0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
pr = cProfile.Profile(); pr.enable()
for _ in range(50): tuner.step(batch, label, epoch=0)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
out = f"batch {B}: host issue time {t_issue / 50 * 1e3:.2f} ms/step, wall {t_all / 50 * 1e3:.2f} ms/step (50 steps)\n" + s.getvalue()
open(os.path.join(ROOT, "gpurun_out", "host_profile.txt"), "w").write(out)
print(out[:6000])

# ---- issue-time split (host-bound regime: wall == issue): forward / backward / optimizer, no profiler attached
import types
fw = bw = op = 0.0
orig_backward = torch.Tensor.backward
def timed_backward(self, *a, **k):
    global bw
    t = time.perf_counter(); r = orig_backward(self, *a, **k); bw += time.perf_counter() - t; return r
torch.Tensor.backward = timed_backward
orig_opt = tuner.optimizer_step
def timed_opt(*a, **k):
    global op
    t = time.perf_counter(); r = orig_opt(*a, **k); op += time.perf_counter() - t; return r
tuner.optimizer_step = timed_opt
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): tuner.step(batch, label, epoch=0)
tot = time.perf_counter() - t0
torch.cuda.synchronize()
line = f"issue split per step: total {tot / 50 * 1e3:.2f} ms = forward+loss {(tot - bw - op) / 50 * 1e3:.2f} + backward {bw / 50 * 1e3:.2f} + optimizer {op / 50 * 1e3:.2f}"
print(line)
open(os.path.join(ROOT, "gpurun_out", "host_profile.txt"), "a").write(line + "\n")

# ---- the backward runs on autograd's device thread: profile THAT thread (a cProfile enabled from inside the first backward node)
import threading
from mmdti_hip import functional as Fn
bw_prof = {}
orig_bw = Fn.CELossFn.backward
def hooked(ctx, *a):
    tid = threading.get_ident()
    if tid not in bw_prof:
        bw_prof[tid] = cProfile.Profile(); bw_prof[tid].enable()
    return orig_bw(ctx, *a)
Fn.CELossFn.backward = staticmethod(hooked)
torch.Tensor.backward = orig_backward
for _ in range(50): tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
for tid, prf in bw_prof.items():
    s2 = io.StringIO()
    try:
        prf.create_stats(); pstats.Stats(prf, stream=s2).sort_stats("tottime").print_stats(40)
    except Exception as e:
        s2.write(f"(could not read the backward thread's profile: {e})")
    txt = "\n==== backward thread, 50 steps ====\n" + s2.getvalue()
    print(txt[:7000])
    open(os.path.join(ROOT, "gpurun_out", "host_profile.txt"), "a").write(txt)
