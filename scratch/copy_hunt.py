"""Where do the ~115 device-to-device copy kernels of a small-batch step come from?  torch profiler over 3 steps, grouped by the
Python frame that issued each copy / fill."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(3): tuner.step(batch, label, epoch=0)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add", "aten::mul", "aten::add_", "aten::mul_", "aten::index", "aten::sum", "aten::to", "aten::_to_copy", "aten::zeros", "aten::ones", "aten::arange", "aten::masked_fill", "aten::eq", "aten::ne"):
        st = [f for f in (ev.stack or []) if "mm-dti_amd" in f or "bench" in f or "tasks" in f]
        key = (ev.name, st[0].split("/")[-1] if st else "?", str(ev.input_shapes)[:60])
        cnt[key] += 1
out = [f"{n / 3:6.1f}/step  {k[0]:18s} {k[1]:70s} {k[2]}" for k, n in cnt.most_common(70)]
open(os.path.join(ROOT, "gpurun_out", "copy_hunt.txt"), "w").write("\n".join(out))
print("\n".join(out))
