"""Host enqueue time per train step (single-GPU model), measured with an empty GPU queue ahead.
  python tools/hosttime_single.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
w, kw = make_world(sys.argv[1] if len(sys.argv) > 1 else "cfg3"); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
bs = [m.device_batch(w.batch(B, i)) for i in range(3)]
for i in range(4): m.train_async(bs[i % 3], 1e-3, 1e-4)
torch.cuda.synchronize()
for n in (1, 3, 6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): m.train_async(bs[i % 3], 1e-3, 1e-4)
    host = time.perf_counter() - t
    torch.cuda.synchronize(); wall = time.perf_counter() - t
    print("steps %d: host %.3f ms/step, wall %.3f ms/step" % (n, host / n * 1e3, wall / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): m.train_async(bs[i % 3], 1e-3, 1e-4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
