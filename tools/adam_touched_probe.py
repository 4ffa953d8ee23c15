"""Where the touched-row update's time goes at a small-row shape (CCMR: 5.4 M rows of 16 floats, ~7,600 rows hit per batch):
the scan of the state bytes or the rows?   python tools/adam_touched_probe.py [n_rows] [D] [hits]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5405586
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 7600
lib = _lib.load()
dev = torch.device("cuda:0")
p = torch.randn((N, D), device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p); g = torch.randn_like(p)
flags = torch.ones((N,), dtype=torch.uint8, device=dev)
step = torch.zeros((N,), dtype=torch.int32, device=dev)
ring = torch.zeros((_lib.ADAM_RING + 1,), dtype=torch.float32, device=dev)
T = _lib.AdamTable(p=p.data_ptr(), m=m.data_ptr(), v=v.data_ptr(), g=g.data_ptr(), n_rows=N, D=D, row_flags=flags.data_ptr(),
                   row_step=step.data_ptr(), alpha_ring=ring.data_ptr(), beta1=0.9, beta2=0.999, eps=1e-8, id_status=None,
                   skipped_steps=None)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
evict = torch.zeros((256 << 20,), dtype=torch.float32, device=dev)


def timed(fn, prep, n=30):
    tot = 0.0
    for i in range(n + 3):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i + 1); b.record(); b.synchronize()
        if i >= 3:
            tot += a.elapsed_time(b)
    return tot / n * 1e3


for hits in (0, H // 8, H, 8 * H):
    rows = torch.randperm(N, device=dev)[:max(hits, 1)].to(torch.int32).sort().values
    fresh = len(sys.argv) > 4            # a different set of rows every call (cold rows, cold translations), as in a training run
    def prep():
        global rows
        flags.fill_(1)
        if hits:
            if fresh:
                rows = torch.randint(0, N, (hits,), device=dev).unique().to(torch.int32)
            flags[rows.long()] = 2
        if fresh:
            evict.add_(1.0)           # 1 GB through the caches: the state bytes and the rows come from HBM, as after a step's other traffic
        torch.cuda.synchronize()
    t_scan = timed(lambda st: _lib.check(lib.score_adam_touched(C.byref(T), st, 1e-3, s), "touched"), prep)
    nrows = torch.tensor([hits], dtype=torch.int32, device=dev)
    def by_list(st):
        nrows.fill_(rows.numel())
        _lib.check(lib.score_adam_touched_rows(C.byref(T), rows.data_ptr(), nrows.data_ptr(), max(hits, 1), st, 1e-3, s), "rows")
    t_list = timed(by_list, prep) if hits else float("nan")
    print("n_rows %d D %d hits %6d: state-byte scan form %6.1f us   row-list form %6.1f us" % (N, D, hits, t_scan, t_list))
