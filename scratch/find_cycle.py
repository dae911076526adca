import sys, gc, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import bench
from mmdti_hip.trainer import FineTuner
model,_=bench.build_model(); model=model.cuda().train()
tuner=FineTuner(model,"classification",total_steps=10000)
_,batch,label=bench.synth(8,20,24,seed=1)
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda()
tuner.step(batch,label,epoch=0)
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
tuner.step(batch,label,epoch=0)
n=gc.collect()
print('collected',n,'garbage',len(gc.garbage))
from collections import Counter
print(Counter(type(o).__name__ for o in gc.garbage).most_common(15))
tens=[o for o in gc.garbage if isinstance(o,torch.Tensor)]
print('tensors in garbage',len(tens), sum(t.numel()*t.element_size() for t in tens)/2**20,'MiB')
for o in gc.garbage:
    if type(o).__name__.endswith('Backward') or 'Fn' in type(o).__name__:
        print('node', type(o).__name__)
# find which objects refer to the largest tensor
if tens:
    big=max(tens,key=lambda t:t.numel())
    print('big',tuple(big.shape),big.dtype, big.grad_fn)
    for r in gc.get_referrers(big)[:6]:
        print('  ref by', type(r).__name__, (list(r.keys())[:12] if isinstance(r,dict) else ''))
