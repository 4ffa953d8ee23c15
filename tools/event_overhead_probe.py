import torch
torch.cuda.set_device(0)
x=torch.zeros(1<<20,device='cuda')
ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for i in range(3): x.add_(1)
torch.cuda.synchronize()
for a,b in ev:
    x.add_(1); a.record(); b.record(); x.add_(1)
torch.cuda.synchronize()
import numpy as np
t=[a.elapsed_time(b) for a,b in ev]
print('empty pair ms: mean %.5f min %.5f max %.5f'%(np.mean(t),min(t),max(t)))
