"""Host enqueue time per train step at a small shape for several debug_flags (e.g. 32: library sort, 256: sort.hip).
  python tools/hosttime_flags.py [config] [flags ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
w, kw = make_world(sys.argv[1] if len(sys.argv) > 1 else "tmall_default"); B = kw.pop("batch")
flags = [int(x) for x in sys.argv[2:]] or [32, 256]
m = SCORE(seed=1, **kw)
bs = [m.device_batch(w.batch(B, i)) for i in range(4)]
for rep in range(3):
    for f in flags:
        m.debug_flags = f
        for i in range(6): m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
        torch.cuda.synchronize()
        n = 200
        t = time.perf_counter()
        for i in range(n): m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
        host = time.perf_counter() - t
        torch.cuda.synchronize(); wall = time.perf_counter() - t
        print("flags %3d: host %.4f ms/step, wall %.4f ms/step" % (f, host / n * 1e3, wall / n * 1e3), flush=True)
# host time of the plan call alone
import ctypes as C
from score_amd import _lib
for f in flags:
    m.debug_flags = f
    db = bs[0]
    lay, ws = m._workspace(db.B)
    st = m._state(ws)
    for _ in range(5): m.lib.score_index_plan(C.byref(m.cfg), C.byref(st), C.byref(db.struct), 1, 0, m._stream())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): m.lib.score_index_plan(C.byref(m.cfg), C.byref(st), C.byref(db.struct), 1, 0, m._stream())
    host = time.perf_counter() - t
    torch.cuda.synchronize(); wall = time.perf_counter() - t
    print("flags %3d: score_index_plan host %.1f us per call, device %.1f us" % (f, host / 200 * 1e6, wall / 200 * 1e6), flush=True)
