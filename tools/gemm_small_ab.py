"""A/B of the f32 kernel's split-K thresholds on the path's small (batch-row) GEMMs: builds gemm.hip +
gemm_bf16x3.hip into stand-alone libraries with different -DGEMM_SPLITK_* and times score_gemm."""
import ctypes as C, os, subprocess, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(); libs = {}
variants = [("k512/128", []), ("k192/64", ["-DGEMM_SPLITK_MIN_K=192", "-DGEMM_SPLITK_MIN_CHUNK=64"]),
            ("k128/32", ["-DGEMM_SPLITK_MIN_K=128", "-DGEMM_SPLITK_MIN_CHUNK=32"])]
for name, defs in variants:
    so = os.path.join(tmp, "g_%d.so" % len(libs))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "score_amd", "csrc", "gemm.hip"),
                           os.path.join(root, "score_amd", "csrc", "gemm_bf16x3.hip"), "-o", so])
    libs[name] = C.CDLL(so)
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(0, 1024, 296, 448), (0, 1024, 80, 296), (0, 20480, 40, 80), (0, 1024, 200, 704), (0, 1024, 80, 200),
          (1, 1024, 200, 80), (1, 1024, 704, 200), (1, 20480, 80, 40), (1, 1024, 296, 80), (1, 1024, 448, 296)]
scratch = torch.empty((1 << 22,), device="cuda")
tot = {n: 0.0 for n, _ in variants}
for tr, M, N, K in shapes:
    a = torch.randn((M, K), device="cuda"); b = torch.randn((K, N), device="cuda")
    A = a if tr != 2 else a.t().contiguous(); Bm = b if tr != 1 else b.t().contiguous()
    c = torch.empty((M, N), device="cuda")
    out = []
    for name, _ in variants:
        lib = libs[name]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: lib.score_gemm(tr, M, N, K, P(A), A.shape[1], P(Bm), Bm.shape[1], P(c), N, None, 16,
                                      C.c_float(1.0), None, C.c_uint64(0), P(scratch), C.c_int64(scratch.numel()), st)
        call(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        tot[name] += best
        out.append("%s %6.1f us" % (name, best * 1e3))
    print("trans=%d M=%d N=%d K=%d: %s" % (tr, M, N, K, "  ".join(out)))
print("sum:", "  ".join("%s %.1f us" % (n, t * 1e3) for n, t in tot.items()))
