"""Which Python call sites issue the device copies / fills / small aten kernels of a small-batch step: the candidate tensor methods are
wrapped and counted per (method, caller line) over one step."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import packing_fields, atom_counts
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, _ = bench.build_model(); model = model.cuda().train()
tuner = FineTuner(model, "classification", total_steps=10000)
_, batch, label = bench.synth(B, 128, 256, seed=1234, ragged=True)
host = dict(packing_fields(batch), atom_counts=atom_counts(batch["src_tokens"], 0))
batch = {k: v.cuda() for k, v in batch.items()}; label = label.cuda(); batch.update(host)
for _ in range(5): tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
cnt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack(limit=12)[:-2]):
        if "mm-dti_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def wrap(obj, name):
    real = getattr(obj, name)
    def f(*a, **k):
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        if t is None or t.is_cuda or name in ("zeros", "ones", "full", "arange", "tensor", "cat", "stack"):
            cnt[(name, site())] += 1
        return real(*a, **k)
    setattr(obj, name, f)
def wrap_h2d(name):
    real = getattr(torch.Tensor, name)
    def f(*a, **k):
        t = a[0]
        if isinstance(t, torch.Tensor) and not t.is_cuda:
            r = real(*a, **k)
            if isinstance(r, torch.Tensor) and r.is_cuda:
                cnt[("H2D " + name + f" {tuple(t.shape)}", site())] += 1
            return r
        return real(*a, **k)
    setattr(torch.Tensor, name, f)
for n in ("clone", "contiguous", "copy_", "to", "float", "long", "int", "bool", "masked_fill_", "masked_fill", "fill_", "zero_", "add_", "mul_", "__getitem__", "sum", "eq", "ne"):
    wrap(torch.Tensor, n)
for n in ("cat", "stack", "zeros", "ones", "full", "arange", "zeros_like", "ones_like", "as_tensor", "tensor"):
    wrap(torch, n)
wrap_h2d("to"); wrap_h2d("cuda")
tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
out = [f"{n:5d}  {k[0]:14s} {k[1]}" for k, n in cnt.most_common(60)]
open(os.path.join(ROOT, "gpurun_out", "copy_hunt.txt"), "w").write("\n".join(out))
print("\n".join(out))
