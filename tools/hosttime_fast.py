"""Host time of the one-call step (model.train_async with the next batch announced): inside score_train_step vs the Python around it.
  python tools/hosttime_fast.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
cfg = sys.argv[1] if len(sys.argv) > 1 else "tmall_default"
w, kw = make_world(cfg); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
bs = [m.device_batch(w.batch(B, i)) for i in range(8)]
for i in range(200): m.train_async(bs[i % 8], 1e-3, 1e-4, next_batch=bs[(i + 1) % 8])
torch.cuda.synchronize()
inner = [0.0, 0]
orig = m.lib.score_train_step
def timed(*a):
    t = time.perf_counter(); r = orig(*a); inner[0] += time.perf_counter() - t; inner[1] += 1; return r
m.lib.score_train_step = timed
N = 2000
t = time.perf_counter()
for i in range(N): m.train_async(bs[i % 8], 1e-3, 1e-4, next_batch=bs[(i + 1) % 8])
host = time.perf_counter() - t
torch.cuda.synchronize(); wall = time.perf_counter() - t
print("%s: host %.1f us/step (inside score_train_step %.1f us over %d calls), wall %.1f us/step" % (cfg, host / N * 1e6, inner[0] / max(inner[1], 1) * 1e6, inner[1], wall / N * 1e6))
m.lib.score_train_step = orig
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(300): m.train_async(bs[i % 8], 1e-3, 1e-4, next_batch=bs[(i + 1) % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
