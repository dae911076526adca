import os, sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
import bench
from mmdti_hip.trainer import FineTuner
model,_=bench.build_model(); model=model.cuda().train()
tuner=FineTuner(model,"classification",total_steps=400,learning_rate=1e-4)
RAG=os.environ.get('SOAK_RAGGED')=='1'
_,batch,label=bench.synth(256,128,256,seed=1234,ragged=RAG)
if RAG:
    from mmdti_hip.collate import atom_counts
    counts=atom_counts(batch['src_tokens'],0)
batch={k:v.cuda() for k,v in batch.items()}; label=label.cuda()
if RAG: batch['atom_counts']=counts
ts=[]; losses=[]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    out=tuner.step(batch,label,epoch=0)
    torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    if i%50==0:
        losses.append((i,float(out.loss),float(out.task_loss),float(out.infonce_loss),float(out.ct_loss)))
import statistics
print('step ms: median %.1f p95 %.1f max(after 5) %.1f'%(statistics.median(ts[5:]), sorted(ts[5:])[int(0.95*len(ts[5:]))], max(ts[5:])))
for l in losses: print('step %3d loss %.4f task %.4f infonce %.4f ct %.4f'%l)
print('finite params:', all(torch.isfinite(p).all().item() for p in model.parameters()))
print('mem GB alloc %.1f peak %.1f reserved %.1f'%(torch.cuda.memory_allocated()/2**30, torch.cuda.max_memory_allocated()/2**30, torch.cuda.memory_reserved()/2**30))
