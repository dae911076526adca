import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mm-dti_amd')
from mmdti_hip import ops
g=dict(np.load('/root/repo/tests/golden/g4_fds_gauss51.npz'))
bn,bs=int(g['cfg_bucket_num']),int(g['cfg_bucket_start']); mn,bw=float(g['min_value']),float(g['bin_width'])
lab=torch.from_numpy(g['labels'])[:,0].contiguous().cuda()
bins,flags=ops.fds_bins(lab,mn,bw,bs,bn)
nb,D=bn-bs,16
rm,rv,tr=torch.zeros(nb,D,device='cuda'),torch.ones(nb,D,device='cuda'),torch.zeros(nb,device='cuda')
ops.fds_update_stats(torch.from_numpy(g['feats0']).cuda(),bins,flags,bs,bn,0.0,rm,rv,tr)
print('rv col3',rv[:,3].tolist()); print('rm col3',rm[:,3].tolist())
win=torch.from_numpy(g['window']).cuda()
sm,sv=ops.fds_smooth_stats(rm,win),ops.fds_smooth_stats(rv,win)
xb=torch.from_numpy(g['xb']).cuda()
b40,f40=ops.fds_bins(lab[:40].contiguous(),mn,bw,bs,bn)
y,sc=ops.fds_smooth(xb,b40,f40,bs,bn,rm,rv,sm,sv)
print('y-x col3',(y-xb)[:8,3].tolist()); print('sc col3',sc[:8,3].tolist())
print('flags',f40.tolist(), b40.tolist())
