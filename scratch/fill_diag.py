"""Which Python line launches the large fill kernels of a step?  (TorchDispatchMode: catches them in the autograd thread too.)"""
import os, sys, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import bench
from mmdti_hip.trainer import FineTuner
from torch.utils._python_dispatch import TorchDispatchMode
model, _ = bench.build_model(); model = model.cuda().train()
tuner = FineTuner(model, "classification", total_steps=100)
_, batch, label = bench.synth(64, 128, 256, seed=1)
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda()
for _ in range(2): tuner.step(batch, label)
torch.cuda.synchronize()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("zero", "fill", "zeros", "full", "add", "copy", "_to_copy")):
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.numel() > 500_000:
                fr = [f"{f.filename.split('/')[-1]}:{f.lineno}:{f.name}" for f in traceback.extract_stack() if "mmdti_hip" in f.filename or "bench.py" in f.filename]
                print(name, tuple(t.shape), t.dtype, "<-", " / ".join(fr[-4:]) or "(autograd engine)")
        return out
with Spy():
    tuner.step(batch, label)
torch.cuda.synchronize()
