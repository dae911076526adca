import sys, types, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd'); sys.path.insert(0,'/root/repo/tests')
from mmdti_hip import ops
import functools
def wrap(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out,(tuple,list)) else (out,)
        for i,o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel()>0:
                if torch.isnan(o.float()).any():
                    print("NaN in output", i, "of", name, tuple(o.shape), "inputs:", [tuple(x.shape) for x in a if isinstance(x,torch.Tensor)], {kk:vv for kk,vv in k.items() if not isinstance(vv,torch.Tensor)})
        return out
    return w
for n in dir(ops):
    f=getattr(ops,n)
    if isinstance(f, types.FunctionType) and not n.startswith('_') and n not in ('lib',):
        setattr(ops, n, wrap(n,f))
import test_modules_gpu as tm
from types import SimpleNamespace
import mmdti_hip.models.transformers as tr, mmdti_hip.models.infonce as inf, mmdti_hip.models.contrastive as ct, mmdti_hip.models.fds as fds, mmdti_hip.models.bert_layers as bl, mmdti_hip.models.mm_model as mm
M=SimpleNamespace(tr=tr,inf=inf,ct=ct,fds=fds,bl=bl,mm=mm)
from oracle import mmdti_oracle as O
for rep in range(3):
    torch.empty(1<<28, device='cuda').fill_(float('nan'))  # poison the allocator cache
    torch.cuda.empty_cache()
    x = torch.empty(1<<26, device='cuda').fill_(float('nan')); del x
    ocfg,P,model = tm._tiny_model(M,'classification',2)
    batch,label = O.synth_batch(6,10,14,ocfg,seed=5,ragged=True)
    model.eval()
    dev={k:v.cuda() for k,v in batch.items()}
    logits,infonce,ctl = model(**dev, return_infonce_loss=True, return_ct_loss=True, net_target=label.cuda())
    print(rep, 'logits nan', torch.isnan(logits).any().item(), float(infonce), float(ctl))
