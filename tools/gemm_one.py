"""Run score_gemm on one shape a few times (for rocprofv3 counter passes):
python3 tools/gemm_one.py TRANS M N K [FLAGS=16] [REPS=5]"""
import sys; sys.path.insert(0, '.')
import ctypes as C, torch
from score_amd import _lib
lib = _lib.load()
tr, M, N, K = (int(x) for x in sys.argv[1:5])
FL = int(sys.argv[5]) if len(sys.argv) > 5 else 16
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
P = lambda t: C.c_void_p(t.data_ptr())
a = torch.randn((M, K), device='cuda'); b = torch.randn((K, N), device='cuda')
A = a if tr != 2 else a.t().contiguous(); Bm = b if tr != 1 else b.t().contiguous()
c = torch.empty((M, N), device='cuda'); scratch = torch.empty((1 << 22,), device='cuda')
for _ in range(reps):
    lib.score_gemm(tr, M, N, K, P(A), A.shape[1], P(Bm), Bm.shape[1], P(c), N, None, FL, 1.0, None, 0, P(scratch),
                   scratch.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
