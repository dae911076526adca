import os, sys
os.environ["MMDTI_FORCE_DDP"]="1"; os.environ.setdefault("MASTER_PORT","29577")
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import torch, bench
from mmdti_hip.parallel import init_from_env
from mmdti_hip.trainer import FineTuner
init_from_env(force=True)
model,_=bench.build_model(); model=model.cuda().train()
cfg,batch,label=bench.synth(64,128,256,1)
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda()
t=FineTuner(model,"classification",distributed=True)
t.step(batch,label); torch.cuda.synchronize()
names={id(p):n for n,p in model.named_parameters()}
print("overlapped",t.reducer.overlapped,"of",len(t.reducer.buckets))
for b,ids in enumerate(t.reducer.unreported()):
    print(b,t.reducer.buckets[b],[names[i] for i in ids])
order=[names[id(p)] for p in t.arena.params]
print(order[:3],order[-3:])
torch.distributed.destroy_process_group()
