"""gemm_x3w (weights pre-split, register-only) against the LDS-staged bf16x3 kernel and the f32-MFMA kernel on the
path's two largest products.  Run on the GPU box: python tools/x3w_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess, tempfile
from score_amd import _lib
lib = _lib.load()
# (round 3: gemm_x3w.hip lives under tools/x3w/, outside the product library: built here through tools/x3w_wrap.hip)
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_so = os.path.join(tempfile.mkdtemp(), "x3w.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       "-I" + os.path.join(_root, "include"), "-I" + os.path.join(_root, "score_amd", "csrc"),
                       os.path.join(_root, "tools", "x3w_wrap.hip"), "-o", _so])
xlib = C.CDLL(_so)
xlib.score_gemm_weights_scratch_floats.restype = C.c_int64
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
for trans, M, N, K in [(0, 20480, 384, 448), (1, 20480, 448, 384), (0, 18432, 384, 448), (0, 196608, 768, 384), (1, 196608, 384, 768)]:
    a = torch.randn((M, K), device="cuda"); w = torch.randn((K, N) if trans == 0 else (N, K), device="cuda")
    c = torch.empty((M, N), device="cuda"); bias = torch.randn((N,), device="cuda")
    scr = torch.empty((int(xlib.score_gemm_weights_scratch_floats(N, K)),), device="cuda")
    scr2 = torch.empty((1 << 22,), device="cuda")
    t_w = timeit(lambda: xlib.score_gemm_weights(trans, M, N, K, P(a), K, P(w), w.shape[1], P(c), N, P(bias), 1, P(scr), C.c_int64(scr.numel()), st()))
    t_x3 = timeit(lambda: lib.score_gemm(trans, M, N, K, P(a), K, P(w), w.shape[1], P(c), N, P(bias), 1 | 32, 1.0, None, 0, P(scr2), scr2.numel(), st()))
    fl = 2.0 * M * N * K
    print("trans=%d M=%d N=%d K=%d: x3w %7.1f us (%5.1f TF-eq, incl. the fragment build)   bf16x3 %7.1f us (%5.1f TF-eq)" %
          (trans, M, N, K, t_w, fl / t_w / 1e6, t_x3, fl / t_x3 / 1e6), flush=True)
