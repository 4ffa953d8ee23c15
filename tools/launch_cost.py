"""Host cost of one kernel launch through libscore_hip.so (score_gather_fwd on a tiny input, asynchronous): for A/B of builds
   python tools/launch_cost.py <lib.so> [<lib2.so> ...]"""
import ctypes as C, sys, time, torch
x = torch.zeros((64, 16), device="cuda"); idx = torch.zeros((8,), dtype=torch.int32, device="cuda"); out = torch.empty((8, 16), device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = [(p, C.CDLL(p)) for p in sys.argv[1:]]
for rep in range(3):
    for p, lib in libs:
        f = lib.score_gather_fwd
        f.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        args = (C.c_void_p(x.data_ptr()), 64, 16, C.c_void_p(idx.data_ptr()), 8, C.c_void_p(out.data_ptr()), st)
        for _ in range(200): f(*args)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2000): f(*args)
        dt = time.perf_counter() - t
        torch.cuda.synchronize()
        print("%-60s %.2f us per launch (host)" % (p[-60:], dt / 2000 * 1e6), flush=True)
