import os,sys,time
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from score_amd.synth import make_world
from score_amd.dist import ShardedSCORE
w,kw=make_world("cfg3"); B=kw.pop("batch")
m=ShardedSCORE(seed=1,**kw)
bs=[m.device_batch(w.batch(B,i)) for i in range(3)]
for i in range(4): m.train_async(bs[i%3],1e-3,1e-4)
torch.cuda.synchronize()
T={"fb":0,"adam":0,"pre":0}; N=20
t_all=time.perf_counter()
for i in range(N):
    t=time.perf_counter(); m.forward_backward(bs[i%3],1e-4,0.8,None,bs[(i+1)%3]); T["fb"]+=time.perf_counter()-t
    t=time.perf_counter(); m.apply_adam(1e-3,1e-4); T["adam"]+=time.perf_counter()-t
host=time.perf_counter()-t_all
torch.cuda.synchronize(); wall=time.perf_counter()-t_all
print("host ms/step", host/N*1e3, "wall ms/step", wall/N*1e3, {k:v/N*1e3 for k,v in T.items()})
# break down forward_backward host time
import cProfile,pstats
pr=cProfile.Profile(); pr.enable()
for i in range(10):
    m.forward_backward(bs[i%3],1e-4,0.8,None,bs[(i+1)%3]); m.apply_adam(1e-3,1e-4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
dist.destroy_process_group()
