"""Host time of one training step by C-ABI call (the small shapes are bound by what the host spends enqueueing):
    python tools/host_calls.py [config] [debug_flags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE

w, kw = make_world(sys.argv[1] if len(sys.argv) > 1 else "tmall_default"); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
m.debug_flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bs = [m.device_batch(w.batch(B, i)) for i in range(4)]


class Timed(object):
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib); object.__setattr__(self, "t", {}); object.__setattr__(self, "n", {})

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
            self.n[name] = self.n.get(name, 0) + 1
            return r
        return call


for i in range(20):
    m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
torch.cuda.synchronize()
tl = Timed(m.lib)
m.lib = tl
n = 300
t0 = time.perf_counter()
for i in range(n):
    m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("host %.1f us/step, wall %.1f us/step" % (host / n * 1e6, wall / n * 1e6))
tot = 0.0
for k in sorted(tl.t, key=lambda k: -tl.t[k]):
    print("  %-36s %6.1f us/step (%.1f calls)" % (k, tl.t[k] / n * 1e6, tl.n[k] / n))
    tot += tl.t[k]
print("  %-36s %6.1f us/step" % ("python + torch (events, streams)", (host - tot) / n * 1e6))
