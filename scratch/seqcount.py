import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import packing_fields, atom_counts
from mmdti_hip import functional as Fn
B=int(sys.argv[1])
model,_=bench.build_model(); model=model.cuda().train()
tuner=FineTuner(model,"classification",total_steps=10000)
_,batch,label=bench.synth(B,128,256,seed=1234,ragged=True)
host=dict(packing_fields(batch),atom_counts=atom_counts(batch["src_tokens"],0))
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda(); batch.update(host)
cnt={}
for name in ("_unimol_layer_fwd_seq","_unimol_layer_bwd_seq","_bert_layer_fwd_seq","_bert_layer_bwd_seq"):
    real=getattr(Fn,name)
    def mk(real,name):
        def f(*a,**k):
            cnt[name]=cnt.get(name,0)+1
            return real(*a,**k)
        return f
    setattr(Fn,name,mk(real,name))
tuner.step(batch,label,epoch=0); torch.cuda.synchronize()
print(B, model.last_layout, cnt, [ (p.M if p is not None else None) for p in (model._pack_cache[3] if model._pack_cache else (None,None))])
