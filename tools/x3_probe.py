"""Where does gemm_bf16x3 spend its time?  Builds the kernel three ways (as shipped / split stripped /
MFMA stripped) and times each on the path's GEMM shapes.  Run on the GPU box: python tools/x3_probe.py"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("full", []), ("setprio1", ["-DX3_SETPRIO=1"]), ("setprio3", ["-DX3_SETPRIO=3"]), ("nosplit", ["-DX3_PROBE_NOSPLIT"]), ("nomfma", ["-DX3_PROBE_NOMFMA"]),
            ("neither", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA"]),
            ("n-noA", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA"]),
            ("n-noB", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADB"]),
            ("n-noAB", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB"]),
            ("n-nostore", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOSTORE"]),
            ("n-noABst", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB",
                          "-DX3_PROBE_NOSTORE"]),
            ("n-noABst-noldsw", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB",
                          "-DX3_PROBE_NOSTORE", "-DX3_PROBE_NOLDSW"]),
            ("n-noABst-noldsr", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB",
                          "-DX3_PROBE_NOSTORE", "-DX3_PROBE_NOLDSR"]),
            ("n-noABst-nolds", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOMFMA", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB",
                          "-DX3_PROBE_NOSTORE", "-DX3_PROBE_NOLDSR", "-DX3_PROBE_NOLDSW"]),
            ("full-noldsw", ["-DX3_PROBE_NOLDSW"]),
            ("full-noldsr", ["-DX3_PROBE_NOLDSR"]),
            ("mfma-only", ["-DX3_PROBE_NOSPLIT", "-DX3_PROBE_NOLOADA", "-DX3_PROBE_NOLOADB",
                          "-DX3_PROBE_NOSTORE", "-DX3_PROBE_NOLDSR", "-DX3_PROBE_NOLDSW"])]
if len(sys.argv) > 1:       # python tools/x3_probe.py full setprio1 ...
    variants = [v for v in variants if v[0] in sys.argv[1:]]
tmp = tempfile.mkdtemp()
libs = {}
for name, defs in variants:
    so = os.path.join(tmp, "probe_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSCORE_PROBE_BUILD",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "tools", "x3_probe_wrap.hip"), "-o", so])
    libs[name] = C.CDLL(so)
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(0, 2, 20480, 384, 448), (2, 2, 448, 384, 20480)]
if os.environ.get('X3_WM1'):
    shapes = [(0, 1, 20480, 384, 448), (0, 2, 20480, 384, 448), (1, 1, 18432, 448, 384), (1, 2, 18432, 448, 384)]
for tr, wm, M, N, K in shapes:
    a = torch.randn((M, K), device="cuda"); b = torch.randn((K, N), device="cuda")
    A = a if tr != 2 else a.t().contiguous(); Bm = b if tr != 1 else b.t().contiguous()
    c = torch.empty((M, N), device="cuda")
    gx, gy = (N + 127) // 128, (M + 64 * wm - 1) // (64 * wm)
    gz, kc, slab = 1, K, None
    if gx * gy < 256 and K >= 512:
        gz = min((320 + gx * gy - 1) // (gx * gy), K // 128)
        kc = ((K + gz - 1) // gz + 31) // 32 * 32
        gz = (K + kc - 1) // kc
        slab = torch.empty((gz * M * N,), device="cuda")
    out = []
    for name, _ in variants:
        lib = libs[name]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: lib.probe_launch(tr, wm, gx, gy, gz, M, N, K, P(A), A.shape[1], P(Bm), Bm.shape[1], P(c), N, kc,
                                        P(slab) if slab is not None else None, st)
        call(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        out.append("%s %6.1f us" % (name, best * 1e3))
    print("trans=%d wm=%d M=%d N=%d K=%d grid=(%d,%d,%d): %s" % (tr, wm, M, N, K, gx, gy, gz, "  ".join(out)))
