"""Where does the fused attention forward (head_fused.hip, attn_fwd_fused_kernel) spend its time?  Builds it with one
ingredient stripped at a time (results are wrong in those builds: timing only) at the cfg-3 shape (B = 1024, T = 18).
Run on the GPU box: python tools/attn_fused_probe.py [variant ...]"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("full", []), ("nopreload-notail", ["-DAFP_NOPRELOAD", "-DAFP_NOTAIL"]), ("noepi-notail", ["-DAFP_NOEPI", "-DAFP_NOTAIL"]),
            ("nopreload-noepi-nobuild-notail", ["-DAFP_NOPRELOAD", "-DAFP_NOEPI", "-DAFP_NOBUILD", "-DAFP_NOTAIL"]), ("twoacc", ["-DAFP_TWOACC"]), ("twoacc-notail", ["-DAFP_TWOACC", "-DAFP_NOTAIL"]), ("notail", ["-DAFP_NOTAIL"]), ("nomfma", ["-DAFP_NOMFMA"]), ("nobuild", ["-DAFP_NOBUILD"]),
            ("noinp", ["-DAFP_NOINP"]), ("notail-nomfma", ["-DAFP_NOTAIL", "-DAFP_NOMFMA"]),
            ("notail-nomfma-nobuild", ["-DAFP_NOTAIL", "-DAFP_NOMFMA", "-DAFP_NOBUILD"]),
            ("notail-nobuild", ["-DAFP_NOTAIL", "-DAFP_NOBUILD"])]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]]
tmp = tempfile.mkdtemp()
B, T, H, NI = 1024, 18, 128, 40
Dk = 2 * H + NI
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(s, device="cuda", generator=g) * 0.1
q, ur, ir, info = r(B, Dk), r(B * T, H), r(B * T, H), r(B * T, NI)
COPIES = int(os.environ.get("COPIES", "8"))
STRIDE = 2 * Dk * 80 + 48
Weff = r(COPIES, STRIDE)
qz, W4, b4, w5, b5 = r(B, 80), r(80, 40), r(40), r(40), r(1)
length = torch.full((B,), T, dtype=torch.int32, device="cuda")
inp, a1, a2 = torch.empty(B * T, 2 * Dk, device="cuda"), torch.empty(B * T, 80, device="cuda"), torch.empty(B * T, 40, device="cuda")
score, head = torch.empty(B, T, device="cuda"), torch.empty(B, 2 * H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for name, defs in variants:
    so = os.path.join(tmp, "probe_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSCORE_PROBE_BUILD",
                           "-Wno-pass-failed", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "tools", "attn_fused_wrap.hip"), "-o", so])
    lib = C.CDLL(so)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.probe_attn(B, T, H, NI, P(q), P(ur), P(ir), P(info), P(Weff), P(qz), P(W4), P(b4), P(w5), P(b5), P(length),
                                  P(inp), P(a1), P(a2), P(score), P(head), 2 * H, st, COPIES, C.c_int64(STRIDE))
    assert call() == 0
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%-24s %7.3f ms" % (name, best), flush=True)
