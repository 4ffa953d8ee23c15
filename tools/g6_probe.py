import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import score_oracle as so
from test_gpu_adam_tiled import make, G6, NAMES8
z = np.load(G6)
cfg = so.Cfg(*[int(x) for x in z["cfg"]], model_type="SCORE")
P = so.init_params(cfg, int(z["seed"]))
m = make(cfg, 24); m.set_params(P)
gmax = z["gmax/emb_mtx"]; G = gmax.max()
lr_sum = 0
for s, (bi, lr) in enumerate(zip(z["order"], z["lrs"])):
    bt = tuple(z["in%d/%s" % (int(bi), k)] for k in NAMES8)
    m.train(None, bt, float(lr), float(z["reg_lambda"]), keep_prob=1.0); lr_sum += float(lr)
    if s + 1 in (1, 5, 10):
        got = m.table.cpu().numpy(); got[0] = P["emb_mtx"][0]
        d = np.abs(got.astype(np.float64) - z["step%d/emb_mtx" % (s + 1)])
        for lo, hi in ((1e-3, 3e-3), (3e-3, 1e-2), (1e-2, 3e-2), (3e-2, 1e-1), (1e-1, 1.01)):
            sel = (gmax > lo * G) & (gmax <= hi * G)
            if sel.any():
                print(s + 1, "gmax/G in (%g,%g]: n=%d max err %.2e  err*gmax/(G*lr_sum) %.2e" % (lo, hi, sel.sum(), d[sel].max(), (d[sel] * gmax[sel] / G / lr_sum).max()))
