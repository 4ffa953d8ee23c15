"""Shared test helpers: golden loading, oracle config construction."""
import os

import numpy as np

from oracle import score_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ("user_1hop", "user_2hop", "item_1hop", "item_2hop",
         "target_user", "target_item", "label", "length")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, D, H, T, K, Fu, Fi = [int(x) for x in z["cfg"]]
    cfg = so.Cfg(N, D, H, T, K, Fu, Fi, str(z["model_type"]))
    batch = {k[3:]: z[k] for k in z.files if k.startswith("in/") and k[3:] in NAMES}
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    return cfg, params, batch, z


def batch_tuple(batch):
    return tuple(batch[n] for n in NAMES)


def random_batch(rng, cfg, B, zero_frac=0.2):
    c = cfg
    sh = {"user_1hop": (B, c.T, c.K, c.Fi), "user_2hop": (B, c.T, c.K, c.Fu),
          "item_1hop": (B, c.T, c.K, c.Fu), "item_2hop": (B, c.T, c.K, c.Fi),
          "target_user": (B, c.Fu), "target_item": (B, c.Fi)}
    b = {}
    for k, s in sh.items():
        a = rng.integers(1, c.N, s)
        if len(s) == 4:
            a[rng.random(s[:2]) < zero_frac] = 0
        b[k] = a.astype(np.int32)
    b["label"] = rng.integers(0, 2, (B,)).astype(np.int32)
    b["length"] = rng.integers(1, c.T + 1, (B,)).astype(np.int32)
    return b
