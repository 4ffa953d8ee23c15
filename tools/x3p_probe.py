"""Round-3 probe: bf16x3 GEMM from pre-split bf16 planes staged by LDS-DMA (tools/x3p/x3p_probe.hip) against the shipped
LDS-staged kernel with the in-kernel split (score_gemm, flag 32), on C = A . B^T with the path's projection shapes.
Run on the GPU box: python tools/x3p_probe.py"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from score_amd import _lib
lib = _lib.load()
so = os.path.join(tempfile.mkdtemp(), "x3p.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] +
                      [a for a in sys.argv[1:] if a.startswith("-D")] + [os.path.join(root, "tools", "x3p", "x3p_probe.hip"), "-o", so])
x = C.CDLL(so)
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
shapes = [(18432, 384, 448), (20480, 384, 448), (36864, 384, 448), (18432, 448, 384), (196608, 768, 384), (1000, 384, 448)]
if os.environ.get("X3P_SHAPES"):
    shapes = shapes[:int(os.environ["X3P_SHAPES"])]
if os.environ.get("X3P_M"):      # per-CU or aggregate limit?  the same per-block work on fewer / more blocks
    shapes = [(int(m), 384, 448) for m in os.environ["X3P_M"].split(",")]
for M, N, K in shapes:
    a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1
    pa = torch.empty((3, M, K), dtype=torch.int16, device="cuda"); pb = torch.empty((3, N, K), dtype=torch.int16, device="cuda")
    assert x.x3p_split(P(a), C.c_int64(M * K), P(pa), st()) == 0 and x.x3p_split(P(b), C.c_int64(N * K), P(pb), st()) == 0
    c = torch.empty((M, N), device="cuda"); c2 = torch.empty((M, N), device="cuda")
    scr = torch.empty((1 << 22,), device="cuda")
    rc = x.x3p_gemm(P(pa), P(pb), P(c), M, N, K, st()); torch.cuda.synchronize()
    assert rc == 0, rc
    ref = (a.double() @ b.double().t())
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    if any(a_.startswith("-DX3P_NO") for a_ in sys.argv[1:]):
        err = float("nan")
    lib.score_gemm(1, M, N, K, P(a), K, P(b), K, P(c2), N, None, 32, C.c_float(1.0), None, C.c_uint64(0), P(scr), C.c_int64(scr.numel()), st())
    torch.cuda.synchronize()
    err2 = float((c2.double() - ref).abs().max() / ref.abs().max())
    t_p = timeit(lambda: x.x3p_gemm(P(pa), P(pb), P(c), M, N, K, st()))
    t_s = timeit(lambda: lib.score_gemm(1, M, N, K, P(a), K, P(b), K, P(c2), N, None, 32, C.c_float(1.0), None, C.c_uint64(0), P(scr), C.c_int64(scr.numel()), st()))
    t_split = timeit(lambda: x.x3p_split(P(a), C.c_int64(M * K), P(pa), st()))
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d: planes+LDS-DMA %7.1f us (%5.1f TF-eq, err %.1e)   shipped %7.1f us (%5.1f TF-eq, err %.1e)   [standalone split of A: %.1f us]"
          % (M, N, K, t_p, fl / t_p / 1e6, err, t_s, fl / t_s / 1e6, err2, t_split), flush=True)
