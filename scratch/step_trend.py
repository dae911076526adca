import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import bench
from mmdti_hip.trainer import FineTuner
model,_=bench.build_model(); model=model.cuda().train()
tuner=FineTuner(model,"classification",total_steps=10000)
_,batch,label=bench.synth(256,128,256,seed=1234)
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda()
ts=[]
import gc
mode=sys.argv[1] if len(sys.argv)>1 else 'none'
mem=[]
for i in range(80):
    torch.cuda.synchronize(); t0=time.perf_counter()
    tuner.step(batch,label,epoch=0)
    torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    mem.append(torch.cuda.memory_allocated()/2**30)
    if mode=='gc': gc.collect()
print(' '.join(f'{t:.0f}' for t in ts))
print('alloc after step GB:', ' '.join(f'{m:.0f}' for m in mem))
print('gc counts', gc.get_count(), gc.get_threshold())
print('mem GB', torch.cuda.max_memory_allocated()/2**30, torch.cuda.memory_reserved()/2**30)
