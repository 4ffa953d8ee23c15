"""Round-3 probe: bf16x3 GEMM with a whole-N output panel per workgroup (tools/x3n/x3n_probe.hip) against the shipped kernel
(score_gemm, flag 32) on the path's two big products: the GRU input projection of both sides (N = 3H = 384, K = I = 448) and
its input gradient (N = 448, K = 384).  Run on the GPU box: python tools/x3n_probe.py [-DX3N_...]"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from score_amd import _lib
lib = _lib.load()
so = os.path.join(tempfile.mkdtemp(), "x3n.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSCORE_PROBE_BUILD", "-Wno-unused-result",
                       "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] +
                      [a for a in sys.argv[1:] if a.startswith("-D")] + [os.path.join(root, "tools", "x3n", "x3n_probe.hip"), "-o", so])
x = C.CDLL(so)
x.x3n_image_bytes.restype = C.c_int64
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
# (groups, M per group, N, K, trans of the weight operand, mt, nbw)
shapes = [(2, 18432, 384, 448, 1, 9, 3), (2, 20480, 384, 448, 1, 10, 3), (2, 18432, 448, 384, 0, 9, 4), (2, 20480, 448, 384, 0, 10, 4),
          (2, 16384, 384, 448, 1, 8, 3), (4, 36864, 384, 448, 1, 9, 3)]
if os.environ.get("X3N_SHAPES"):
    shapes = shapes[:int(os.environ["X3N_SHAPES"])]
stripped = any(a_.startswith("-DPANEL_PROBE_NO") for a_ in sys.argv[1:])
for G, M, N, K, trans, mt, nbw in shapes:
    a = [torch.randn((M, K), device="cuda") for _ in range(G)]
    b = [(torch.randn((K, N) if trans else (N, K), device="cuda") * 0.1) for _ in range(G)]
    bias = [torch.randn((N,), device="cuda") for _ in range(G)]
    c = [torch.empty((M, N), device="cuda") for _ in range(G)]
    img = [torch.empty((x.x3n_image_bytes(N, K),), dtype=torch.uint8, device="cuda") for _ in range(G)]
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    Aa, Ia, Ca, Ba, Bb = arr(a), arr(img), arr(c), arr(bias), arr(b)
    def prep():
        rc = x.x3n_prep(G, Bb, b[0].shape[1], trans, N, K, Ia, st())
        assert rc == 0, rc
    prep()
    run = lambda: x.x3n_gemm(G, Aa, Ia, Ca, Ba, M, N, K, K, N, mt, st())
    rc = run(); torch.cuda.synchronize()
    assert rc == 0, rc
    err = 0.0
    for g in range(G):
        ref = a[g].double() @ (b[g].double() if trans else b[g].double().t()) + bias[g].double()
        err = max(err, float((c[g].double() - ref).abs().max() / ref.abs().max()))
    if stripped:
        err = float("nan")
    # the shipped kernel on the same rows as ONE product (the engine's grouped launch does the same work)
    Ms = G * M
    a2 = torch.cat(a, 0); c2 = torch.empty((Ms, N), device="cuda"); scr = torch.empty((1 << 22,), device="cuda")
    shipped = lambda: lib.score_gemm(0 if trans else 1, Ms, N, K, P(a2), K, P(b[0]), b[0].shape[1], P(c2), N, P(bias[0]), 32 | 1, C.c_float(1.0), None,
                                     C.c_uint64(0), P(scr), C.c_int64(scr.numel()), st())
    rc = shipped(); torch.cuda.synchronize()
    assert rc == 0, rc
    ref0 = a[0].double() @ (b[0].double() if trans else b[0].double().t()) + bias[0].double()
    err2 = float((c2[:M].double() - ref0).abs().max() / ref0.abs().max())
    t_n = timeit(run); t_s = timeit(shipped); t_p = timeit(prep)
    fl = 2.0 * Ms * N * K
    print("G=%d M=%d N=%d K=%d mt=%d nbw=%d: panel %7.1f us (%5.1f TF-eq, err %.1e)   shipped %7.1f us (%5.1f TF-eq, err %.1e)   [weight images: %.1f us]"
          % (G, M, N, K, mt, nbw, t_n, fl / t_n / 1e6, err, t_s, fl / t_s / 1e6, err2, t_p), flush=True)
