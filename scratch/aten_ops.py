import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import bench
from mmdti_hip.trainer import FineTuner
from torch.profiler import profile, ProfilerActivity
model,_=bench.build_model(); model=model.cuda().train()
tuner=FineTuner(model,"classification",total_steps=400)
_,batch,label=bench.synth(256,128,256,seed=1234)
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda()
for _ in range(3): tuner.step(batch,label)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    tuner.step(batch,label)
    torch.cuda.synchronize()
rows=[(e.key,e.count,e.cpu_time_total) for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda r:-r[1])
for r in rows[:40]: print(f"{r[0]:40s} {r[1]:5d} {r[2]/1e3:8.2f} ms")
