"""Run-to-run spread of the dense table sweep (score_adam_rows, every row live) and its sensitivity to the
relative placement of the p / m / v / g arrays: one allocation, the four arrays `pad` bytes apart beyond their size.
  python tools/adam_layout_probe.py <pad_bytes> [...]      (one fresh process per launch gives a new physical layout)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from score_amd import _lib
lib = _lib.load()
N, D = 1529672, 64
P = lambda t: C.c_void_p(t.data_ptr())
for pad in [int(x) for x in sys.argv[1:]] or [0]:
    n = N * D
    stride = n + pad // 4
    if pad < 0:       # separate allocations (what the model does)
        arrs = [torch.zeros(n, device="cuda") for _ in range(4)]
    else:
        buf = torch.zeros(4 * stride, device="cuda")
        arrs = [buf[i * stride:i * stride + n] for i in range(4)]
    flags = torch.ones(N, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.score_adam_rows(P(arrs[0]), P(arrs[1]), P(arrs[2]), P(arrs[3]), N, D, P(flags), 1e-3, 0.9, 0.999, 1e-8, st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print("pad %8d: %s ms  (%.2f TB/s best)" % (pad, " ".join("%.3f" % t for t in ts), 6 * n * 4 / min(ts) / 1e9))
    del arrs
