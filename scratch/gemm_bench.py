import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import ctypes as C, torch
from score_amd import _lib
lib=_lib.load()
P=lambda t: C.c_void_p(t.data_ptr())
st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes=[(0,20480,256,448),(0,20480,128,448),(0,20480,384,448),(0,20480,80,1184),(0,20480,40,80),
        (1,20480,448,256),(1,20480,448,128),(1,20480,448,384),(1,20480,1184,80),
        (2,448,256,20480),(2,448,384,20480),(2,128,256,20480),(2,1184,80,20480),(2,128,128,20480),(0,1024,200,704)]
scratch=torch.empty((1<<22,),device='cuda')
import sys
FL=int(sys.argv[1]) if len(sys.argv)>1 else 0
for tr,M,N,K in shapes:
    a=torch.randn((M,K),device='cuda'); b=torch.randn((K,N),device='cuda')
    A=a if tr!=2 else a.t().contiguous(); Bm=b if tr!=1 else b.t().contiguous()
    c=torch.empty((M,N),device='cuda')
    for _ in range(3): lib.score_gemm(tr,M,N,K,P(A),A.shape[1],P(Bm),Bm.shape[1],P(c),N,None,FL,1.0,None,0,P(scratch),scratch.numel(),st())
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.score_gemm(tr,M,N,K,P(A),A.shape[1],P(Bm),Bm.shape[1],P(c),N,None,FL,1.0,None,0,P(scratch),scratch.numel(),st())
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    ref=(a.double()@b.double()); err=((c.double()-ref).abs()/( (a.abs().double()@b.abs().double())+1e-30)).max().item()
    print("trans=%d M=%6d N=%5d K=%6d  %8.1f us  %6.1f TF/s  maxerr %.2e"%(tr,M,N,K,ms*1e3,2*M*N*K/ms/1e9,err))
