"""Soak: the product's step (three streams, time-tiled optimizer, look-ahead hints) against the same model with NO second stream
(debug_flags bit 12: everything inline on the launch stream) for many steps on the same batches -- every parameter and the table's
optimizer state bit for bit at the end.  A race between the streams (the publish protocol of csrc/adam_tiled.hip, a missing event)
would show as a difference; the suite's soaks are 600 steps on the small shapes.
  python tools/soak_cfg3.py [config] [steps] [check_every]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from score_amd.model import SCORE
from score_amd.synth import make_world

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
w, kw = make_world(cfg)
B = kw.pop("batch")
a = SCORE(seed=5, **kw)
b = SCORE(seed=5, **kw)
b.debug_flags = 4096
nb = 16
bs_a = [a.device_batch(w.batch(B, 900 + i)) for i in range(nb)]
bs_b = [b.device_batch(w.batch(B, 900 + i)) for i in range(nb)]
t0 = time.time()
bad = 0
for s in range(steps):
    la = a.train_async(bs_a[s % nb], 1e-3, 1e-4, keep_prob=0.8, next_batch=bs_a[(s + 1) % nb])
    lb = b.train_async(bs_b[s % nb], 1e-3, 1e-4, keep_prob=0.8, next_batch=bs_b[(s + 1) % nb])
    if (s + 1) % every == 0 or s + 1 == steps:
        torch.cuda.synchronize()
        same_loss = bool(torch.equal(la, lb))
        pa, pb = a.get_params(), b.get_params()
        diff = [k for k in pa if not np.array_equal(np.asarray(pa[k]), np.asarray(pb[k]))]
        a._flush_adam(); b._flush_adam()
        torch.cuda.synchronize()
        st = bool(torch.equal(a._tbl_m, b._tbl_m) and torch.equal(a._tbl_v, b._tbl_v) and torch.equal(a.w_m, b.w_m) and torch.equal(a.w_v, b.w_v))
        a.check_ids(); b.check_ids()
        print("step %6d  loss %.6f  same loss %s  parameters that differ %s  optimizer state equal %s  (%.0f s)" %
              (s + 1, float(la), same_loss, diff, st, time.time() - t0), flush=True)
        bad += (not same_loss) + len(diff) + (not st)
print("SOAK", "FAILED" if bad else "ok", cfg, steps, "steps")
sys.exit(1 if bad else 0)
