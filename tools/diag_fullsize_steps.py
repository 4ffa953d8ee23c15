"""Diagnostic: where do the HIP model and the CPU restatement part after a few full-size optimizer steps?
(tools/, not a test: prints distributions)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
torch.set_num_threads(16)
from oracle import score_oracle as so
from score_amd.model import SCORE
from score_amd.synth import make_world

w, kw = make_world("cfg3")
B = kw.pop("batch")
STEPS, lr, lam = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1e-3, 1e-4
m = SCORE(seed=17, **kw)
P = m.get_params()
md = SCORE(seed=17, **kw)
md.adam_window = 0                       # the per-step sweep
om = so.OracleModel(kw["feature_size"], kw["eb_dim"], kw["hidden_size"], kw["max_time_len"], kw["obj_per_time_slice"],
                    kw["user_fnum"], kw["item_fnum"], "SCORE", params=P)
bs = [w.batch(B, 40 + i) for i in range(STEPS)]
for i, b in enumerate(bs):
    lg = m.train(None, b, lr, lam, keep_prob=1.0, next_batch=bs[i + 1] if i + 1 < STEPS else None)
    ld = md.train(None, b, lr, lam, keep_prob=1.0)
    lo = om.train(None, b, lr, lam, keep_prob=1.0)
    print(i, lg, ld, lo, abs(lg - lo) / abs(lo), flush=True)
G, Dn, O = m.get_params(), md.get_params(), om.params
print("tiled == sweep bit for bit:", all(np.array_equal(np.asarray(G[k]), np.asarray(Dn[k])) for k in G))
tg, to, t0 = np.asarray(G["emb_mtx"]), np.asarray(O["emb_mtx"]), np.asarray(P["emb_mtx"])
ids = [np.unique(np.concatenate([np.asarray(b[k]).ravel() for k in range(6)])) for b in bs]
cnt = np.zeros(tg.shape[0], dtype=np.int32)
first = np.full(tg.shape[0], -1, dtype=np.int32)
for i, x in enumerate(ids):
    cnt[x] += 1
    first[x[first[x] < 0]] = i
cnt[0] = 0
d = np.abs(tg - to)
for c in range(1, STEPS + 1):
    rows = np.nonzero(cnt == c)[0]
    if rows.size == 0:
        continue
    dd = d[rows]
    print("touched %d times: %7d rows; frac < 2e-5 %.4f  < 2e-4 %.4f  < 1e-3 %.4f  max %.2e" %
          (c, rows.size, (dd < 2e-5).mean(), (dd < 2e-4).mean(), (dd < 1e-3).mean(), dd.max()))
for f in range(STEPS):
    rows = np.nonzero((first == f) & (cnt == 1))[0]
    if rows.size:
        dd = d[rows]
        print("only in step %d: %7d rows; frac < 2e-5 %.4f  < 2e-4 %.4f  max %.2e" % (f, rows.size, (dd < 2e-5).mean(), (dd < 2e-4).mean(), dd.max()))
for k in O:
    if k != "emb_mtx":
        dd = np.abs(np.asarray(G[k]).reshape(np.asarray(O[k]).shape) - np.asarray(O[k]))
        print("%-28s frac < 2e-5 %.4f  < 2e-4 %.4f  max %.2e" % (k, (dd < 2e-5).mean(), (dd < 2e-4).mean(), dd.max()))
