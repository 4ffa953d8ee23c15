"""nested-list walk (score_amd/cext/listpack.c) by native thread count, on a cfg-3 batch as GraphLoader would yield it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from score_amd import _lib
from score_amd.synth import make_world
lp = _lib.listpack()
w, kw = make_world("cfg3"); B = kw.pop("batch")
b = w.batch(B, 3, as_lists=True)
shapes = [np.asarray(x).shape for x in w.batch(B, 3)]
outs = [np.zeros(int(np.prod(s)), dtype=np.int32) for s in shapes]
print("usable cpus", len(os.sched_getaffinity(0)))
for nt in (1, 2, 4, 6, 8, 12, 16):
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        lp.pack_many([(x, o, tuple(s)) for x, o, s in zip(b, outs, shapes)], nt)
        ts.append(time.perf_counter() - t)
    print("threads %2d: %.2f ms per batch (min %.2f)" % (nt, sorted(ts)[3] * 1e3, min(ts) * 1e3))
