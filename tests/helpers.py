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


def away_from_relu_kinks(cfg, params, batch, thr=1e-5, keep_prob=1.0, dropout_masks=None):
    """The batch without the samples that have a relu pre-activation within `thr` of the kink in the oracle's forward pass
    (co-attention scores, dense_3 / dense_4, fc1 / fc2: 10^5 units and more per batch).  A relu network's gradient is
    discontinuous there: two correct fp32 passes whose pre-activation of ONE unit differs in the last bit differ by that
    unit's whole gradient, so a comparison of gradients has to leave such units -- and what depends on them -- out.  Every
    unit belongs to one sample and nothing crosses the batch (oracle.forward "relu_margin_per_sample"), so what depends on a
    unit is its sample: dropping the sample drops exactly that, and the rest of the batch is compared on ARBITRARY inputs
    (no hand-picked seeds).  Returns (batch, dropout_masks or None, indices kept)."""
    import torch
    with torch.no_grad():
        out = so.forward(cfg, so.to_torch_params(params), so.to_torch_batch(batch), keep_prob,
                         [torch.as_tensor(np.asarray(m)) for m in dropout_masks] if dropout_masks is not None else None, 0.0)
    per = np.asarray(out["relu_margin_per_sample"])
    keep = np.nonzero(per >= thr)[0]
    assert keep.size >= max(1, (3 * per.size) // 4), "more than a quarter of the batch sits on a relu kink: %r" % (per,)
    b = {k: np.ascontiguousarray(np.asarray(v)[keep]) for k, v in batch.items()}
    dm = [np.ascontiguousarray(np.asarray(m)[keep]) for m in dropout_masks] if dropout_masks is not None else None
    return b, dm, keep
