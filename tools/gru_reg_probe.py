"""Where does the H = 128 register-resident recurrence (gru.hip) spend its time?  Builds gru.hip with one ingredient
stripped at a time (results are wrong in those builds: timing only) and times both directions at the cfg-3 shape
(B = 1024, T = 18, two GRUs).  Run on the GPU box: python tools/gru_reg_probe.py [variant ...]"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("full", []), ("nomfma", ["-DGRP_NOMFMA"]), ("nostore", ["-DGRP_NOSTORE"]), ("noxload", ["-DGRP_NOXLOAD"]),
            ("nomem", ["-DGRP_NOSTORE", "-DGRP_NOXLOAD"]), ("nomem-nomfma", ["-DGRP_NOSTORE", "-DGRP_NOXLOAD", "-DGRP_NOMFMA"])]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]]
tmp = tempfile.mkdtemp()
B, T, H = 1024, 18, 128
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(s, device="cuda", generator=g) * 0.1
xproj, Wg, Wc = r(2, B * T, 3 * H), r(2, H, 2 * H), r(2, H, H)
length = torch.full((B,), T, dtype=torch.int32, device="cuda")
out, gates = torch.zeros(2, B * T, H, device="cuda"), torch.zeros(2, B * T, 3 * H, device="cuda")
dout, dxproj, rh, hprev = r(2, B * T, H), torch.zeros(2, B * T, 3 * H, device="cuda"), torch.zeros(2, B * T, H, device="cuda"), torch.zeros(2, B * T, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for name, defs in variants:
    so = os.path.join(tmp, "probe_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSCORE_PROBE_BUILD",
                           "-Wno-pass-failed", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "score_amd", "csrc")] + defs +
                          [os.path.join(root, "tools", "gru_reg_wrap.hip"), "-o", so])
    lib = C.CDLL(so)
    res = []
    for d in (0, 1):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: lib.probe_gru(d, B, T, H, P(xproj), P(Wg), P(Wc), P(length), P(out), P(gates), P(dout), P(dxproj),
                                     P(rh), P(hprev), st)
        assert call() == 0
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        res.append(best)
    print("%-13s fwd %7.3f ms   bwd %7.3f ms" % (name, res[0], res[1]), flush=True)
