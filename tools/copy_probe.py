"""float4 stream-copy variants on this box (what bench.py's roofline.peak_measured should use).  python tools/copy_probe.py"""
import ctypes as C, os, subprocess, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(), "probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(root, "tools", "copy_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
names = ["grid-stride U1", "grid-stride U4", "grid-stride U8", "grid-stride U4 nt", "grid-stride U8 nt", "chunk U4", "chunk U8", "hipMemcpyAsync D2D"]
for mib in (1024, 4096):
    n = mib * (1 << 20) // 4
    s = torch.randn(n, device="cuda"); d = torch.empty_like(s)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for which in range(8):
        for blocks in ((0,) if which == 7 else (1024, 2048, 4096, 8192, 16384, 65536) if which < 5 else (2048, 8192, 32768)):
            call = lambda: lib.probe(which, blocks, d.data_ptr(), s.data_ptr(), n, st)
            assert call() == 0
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): call()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            print("%5d MiB %-22s blocks %6d  %8.1f us  %5.2f TB/s" % (mib, names[which], blocks, best * 1e3, 2 * n * 4 / best / 1e9), flush=True)
    del s, d
