import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
w, kw = make_world("tmall_default"); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
bs = [m.device_batch(w.batch(B, i)) for i in range(8)]
for i in range(10):
    m.forward_backward(bs[i % 8], 1e-4, 0.8); m.apply_adam(1e-3, 1e-4)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(300):
    m.forward_backward(bs[i % 8], 1e-4, 0.8); m.apply_adam(1e-3, 1e-4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
