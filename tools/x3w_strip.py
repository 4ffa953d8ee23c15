"""Where does gemm_x3w spend its time?  Builds it with one ingredient stripped at a time (wrong results, timing only).
Run on the GPU box: python tools/x3w_strip.py"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("full", []), ("ra2", ["-DXW_RA=2"]), ("nosplit", ["-DXWP_NOSPLIT"]), ("nomfma", ["-DXWP_NOMFMA"]), ("noA", ["-DXWP_NOLOADA"]),
            ("noB", ["-DXWP_NOLOADB"]), ("noAB", ["-DXWP_NOLOADA", "-DXWP_NOLOADB"]),
            ("mfma-only", ["-DXWP_NOLOADA", "-DXWP_NOLOADB", "-DXWP_NOSPLIT"]),
            ("noAB-nomfma", ["-DXWP_NOLOADA", "-DXWP_NOLOADB", "-DXWP_NOMFMA"])]
tmp = tempfile.mkdtemp()
P = lambda t: C.c_void_p(t.data_ptr())
M, N, K = 20480, 384, 448
a = torch.randn((M, K), device="cuda"); w = torch.randn((K, N), device="cuda"); c = torch.empty((M, N), device="cuda")
bias = torch.randn((N,), device="cuda")
for name, defs in variants:
    so = os.path.join(tmp, "x3w_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "tools", "x3w_wrap.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.score_gemm_weights_scratch_floats.restype = C.c_int64
    scr = torch.empty((int(lib.score_gemm_weights_scratch_floats(N, K)),), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.score_gemm_weights(0, M, N, K, P(a), K, P(w), N, P(c), N, P(bias), 1, P(scr), C.c_int64(scr.numel()), st)
    assert call() == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%-12s %7.1f us" % (name, best * 1e3), flush=True)
