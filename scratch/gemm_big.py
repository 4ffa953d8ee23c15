import sys; sys.path.insert(0,'.')
import ctypes as C, torch
from score_amd import _lib
lib=_lib.load()
P=lambda t: C.c_void_p(t.data_ptr())
st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for FL in (0,16):
  for (tr,M,N,K) in [(0,4096,4096,4096),(0,8192,8192,1024),(0,20480,384,4096),(0,20480,384,448),(0,81920,384,448)]:
    a=torch.randn((M,K),device='cuda'); b=torch.randn((K,N),device='cuda'); c=torch.empty((M,N),device='cuda')
    for _ in range(2): lib.score_gemm(tr,M,N,K,P(a),K,P(b),N,P(c),N,None,FL,1.0,None,0,None,0,st())
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lib.score_gemm(tr,M,N,K,P(a),K,P(b),N,P(c),N,None,FL,1.0,None,0,None,0,st())
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    print("flags=%2d M=%6d N=%5d K=%5d %9.1f us %7.1f TF/s"%(FL,M,N,K,ms*1e3,2*M*N*K/ms/1e9))
