"""cProfile of the host side of a training step at a small shape (where the step is bound by what the host spends):
    python tools/py_profile_step.py [config] [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE

w, kw = make_world(sys.argv[1] if len(sys.argv) > 1 else "tmall_default"); B = kw.pop("batch")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = SCORE(seed=1, **kw)
bs = [m.device_batch(w.batch(B, i)) for i in range(4)]
for i in range(50):
    m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    m.train_async(bs[i % 4], 1e-3, 1e-4, next_batch=bs[(i + 1) % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
