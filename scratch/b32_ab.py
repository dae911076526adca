"""A/B at the reference's default batch (32 molecules, mixed lengths) inside ONE process: blocks of steps alternate between settings
of a module flag.   python scratch/b32_ab.py module.FLAG [rounds]     e.g. functional.SIDE_WGRAD"""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import packing_fields, atom_counts
modname, flag = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("mmdti_hip." + modname)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model, _ = bench.build_model()
model = model.cuda().train()
tuner = FineTuner(model, "classification", total_steps=100000)
_, batch, label = bench.synth(32, 128, 256, seed=8765, ragged=True)
host = dict(packing_fields(batch), atom_counts=atom_counts(batch["src_tokens"], 0))
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda(); batch.update(host)
def block(n=100):
    for _ in range(10): tuner.step(batch, label, epoch=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tuner.step(batch, label, epoch=0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for strict in (None, False):
    model.strict_reference = strict
    res = {True: [], False: []}
    for r in range(rounds):
        for v in (True, False):
            setattr(mod, flag, v)
            res[v].append(block())
    setattr(mod, flag, True)
    print(f"layout {model.last_layout}: {flag}=1 " + " ".join(f"{x:.2f}" for x in res[True]) + f" | {flag}=0 " + " ".join(f"{x:.2f}" for x in res[False]), flush=True)
