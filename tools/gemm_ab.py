import sys; sys.path.insert(0,'.')
import ctypes as C, torch
from score_amd import _lib
lib=_lib.load()
P=lambda t: C.c_void_p(t.data_ptr())
st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes=[(0,20480,80,592),(1,20480,592,80),(2,592,80,20480),(0,20480,256,448),(0,20480,128,448),(0,20480,384,448),(0,20480,80,1184),
        (1,20480,448,256),(1,20480,448,128),(1,20480,448,384),(1,20480,1184,80),
        (2,448,256,20480),(2,448,128,20480),(2,448,384,20480),(2,128,256,20480),(2,1184,80,20480),(2,128,128,20480),(0,1024,200,704),(2,704,200,1024)]
FLB=int(sys.argv[1]) if len(sys.argv)>1 else 16
scratch=torch.empty((1<<22,),device='cuda')
tot=[0,0]
for tr,M,N,K in shapes:
    a=torch.randn((M,K),device='cuda'); b=torch.randn((K,N),device='cuda')
    A=a if tr!=2 else a.t().contiguous(); Bm=b if tr!=1 else b.t().contiguous()
    c=torch.empty((M,N),device='cuda')
    res=[]
    for rnd in range(3):
        for FL in (0,FLB,32):
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            lib.score_gemm(tr,M,N,K,P(A),A.shape[1],P(Bm),Bm.shape[1],P(c),N,None,FL,1.0,None,0,P(scratch),scratch.numel(),st())
            e0.record()
            for _ in range(10): lib.score_gemm(tr,M,N,K,P(A),A.shape[1],P(Bm),Bm.shape[1],P(c),N,None,FL,1.0,None,0,P(scratch),scratch.numel(),st())
            e1.record(); torch.cuda.synchronize()
            res.append((FL,e0.elapsed_time(e1)/10))
    t0=min(t for f,t in res if f==0); t1=min(t for f,t in res if f==FLB); t2=min(t for f,t in res if f==32)
    tot[0]+=t0; tot[1]+=t1
    print("trans=%d M=%6d N=%5d K=%6d  fp32 %7.1f us (%5.1f TF)  x3 %7.1f us (%5.1f TF)  ratio %.2f  forced-x3 %7.1f us (%5.1f TF)"%(tr,M,N,K,t0*1e3,2*M*N*K/t0/1e9,t1*1e3,2*M*N*K/t1/1e9,t0/t1,t2*1e3,2*M*N*K/t2/1e9))
print("sum fp32 %.1f us  x3 %.1f us"%(tot[0]*1e3,tot[1]*1e3))
