"""Would an interleaved [N][4][D] table block (p, m, v, g of a row contiguous: 1 KB) serve the table optimizer's row
kernels better than four separate [N][D] arrays?  Times the touched-rows access pattern of cfg-3 (177 k of 1.53 M rows,
D = 64) in both layouts, and what the forward's gather of p rows would pay for the interleaving.
Run on the GPU box: python tools/adam_layout_probe.py"""
import ctypes as C, os, subprocess, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(), "probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(root, "tools", "adam_layout_wrap.hip"), "-o", so])
lib = C.CDLL(so)
N, D = 1529672, 64
g = torch.Generator(device="cuda").manual_seed(1)
split = torch.rand(4, N, D, device="cuda", generator=g)
inter = torch.rand(N, 4, D, device="cuda", generator=g)
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, n in (("touched rows of a step (177 k, sorted)", 176700), ("rows a batch reads (2.0 M uses, random order)", 2000000)):
    if n < N:
        rows = torch.randperm(N, device="cuda", generator=g)[:n].sort().values.int()
    else:
        rows = torch.randint(0, N, (n,), device="cuda", generator=g).int()
    out = torch.empty(max(n, 200000), D, device="cuda")     # (also the state arrays of variant 4: 1.53 M bytes + 1.53 M words)
    for which, label, byts in ((0, "four arrays   read p,m,v,g write p,m,v", n * D * 4 * 7), (1, "interleaved   read p,m,v,g write p,m,v", n * D * 4 * 7), (4, "four arrays + state byte + step counter", n * D * 4 * 7),
                               (2, "gather p rows, row stride 256 B", n * D * 4 * 2), (3, "gather p rows, row stride 1 KB ", n * D * 4 * 2)):
        a = split if which in (0, 2, 4) else inter
        args = (P(split[0]), P(split[1]), P(split[2]), P(split[3])) if which in (0, 4) else (P(a), None, None, None)
        call = lambda: lib.probe(which, *args, P(rows), n, P(out), st())
        assert call() == 0
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print("%-46s %-42s %7.1f us  %5.2f TB/s" % (name, label, best * 1e3, byts / best / 1e9), flush=True)
