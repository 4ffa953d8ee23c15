"""Why does the H = 128 recurrence take 83 / 98 us inside the cfg-3 step and 47 / 59 us alone (tools/gru_x3_probe.py)?
Times gru_fwd_x3 / gru_bwd_x3 (cfg-3 shape) (a) alone, hot caches; (b) alone, caches flushed in front of every call; (c) beside a
bandwidth hog on another stream (a 1-GiB copy); (d) beside a VALU hog with no memory traffic; (e) beside the library's radix sort
of 2.9 M pairs (what runs beside the forward recurrence in the step).  Run on the GPU box."""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
B, T, H = 1024, 18, 128
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(s, device="cuda", generator=g) * 0.1
xproj, Wg, Wc = r(2, B * T, 3 * H), r(2, H, 2 * H), r(2, H, H)
length = torch.full((B,), T, dtype=torch.int32, device="cuda")
out, gates = torch.zeros(2, B * T, H, device="cuda"), torch.zeros(2, B * T, 3 * H, device="cuda")
dout, dxproj, rh, hprev = r(2, B * T, H), torch.zeros(2, B * T, 3 * H, device="cuda"), torch.zeros(2, B * T, H, device="cuda"), torch.zeros(2, B * T, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
libs = {}
VARIANTS = [("full", []), ("noxload", ["-DXGP_NOXLOAD"]), ("nostore", ["-DXGP_NOSTORE"]), ("nomem", ["-DXGP_NOSTORE", "-DXGP_NOXLOAD"])] + [("pf%d" % k, ["-DXG_PREFETCH=%d" % k]) for k in (0, 2, 4, 6)]
if len(sys.argv) > 1:
    VARIANTS = [v for v in VARIANTS if v[0] in sys.argv[1:]]
for name, defs in VARIANTS:
    so = os.path.join(tmp, "probe_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSCORE_PROBE_BUILD",
                           "-Wno-pass-failed", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "tools", "gru_x3_wrap.hip"), "-o", so])
    libs[name] = C.CDLL(so)
big_a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")   # 1 GiB
big_b = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
keys = torch.randint(0, 1 << 21, (2_900_000,), device="cuda", dtype=torch.int32)
valu_x = torch.randn(256 * 1024, device="cuda")


def hog_bw():
    big_b.copy_(big_a)


def hog_valu():
    y = valu_x
    for _ in range(40):
        y = torch.sin(y) * 1.0001 + 0.5      # small working set: VALU / transcendental bound, L2-resident


def hog_sort():
    torch.sort(keys)


def flush():
    big_b[:64 << 20].fill_(1.0)              # 256 MiB written: past L2 and the memory-side cache


def time_call(lib, d, hog=None, cold=False, n=12):
    main = torch.cuda.current_stream()
    st = C.c_void_p(main.cuda_stream)
    call = lambda: lib.probe_gru(d, B, T, H, P(xproj), P(Wg), P(Wc), P(length), P(out), P(gates), P(dout), P(dxproj), P(rh), P(hprev), st)
    call(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if cold:
            flush()
        torch.cuda.synchronize()
        if hog is not None:
            with torch.cuda.stream(side):
                hog()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name in libs:
    for what, kw in (("alone, hot", {}), ("alone, caches flushed", {"cold": True}), ("beside a 1-GiB copy", {"hog": hog_bw}),
                     ("beside a VALU hog", {"hog": hog_valu}), ("beside torch.sort of 2.9 M keys", {"hog": hog_sort}),
                     ("flushed + beside the sort", {"hog": hog_sort, "cold": True})):
        f, b = time_call(libs[name], 0, **kw), time_call(libs[name], 1, **kw)
        print("%-8s %-32s fwd %6.1f us   bwd %6.1f us" % (name, what, f, b), flush=True)
