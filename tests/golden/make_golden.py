"""Generate the committed golden vectors (G1, G2, G3).

The reference's hot path (code/score/score.py) is a TensorFlow-1.x graph and
cannot be imported or run (no TensorFlow), and the reference holds no golden
vectors of its own, so these fixtures are produced from THIS repo's CPU
restatement (oracle/score_oracle.py, Oracle B fp32, cross-checked here against
the literal fp64 Oracle A).  Parity is therefore "unpinned" by the reference.

Run:  python tests/golden/make_golden.py     (CPU, ~1 min; needs no GPU)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import score_oracle as so          # noqa: E402
from score_amd.synth import make_world         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("user_1hop", "user_2hop", "item_1hop", "item_2hop",
         "target_user", "target_item", "label", "length")


def perturbed_params(cfg, seed):
    """Initial values with non-trivial biases/gamma/beta so every term is exercised."""
    P = so.init_params(cfg, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    for k in P:
        if k != "emb_mtx" and ("bias" in k or "beta" in k or "gamma" in k):
            P[k] = (P[k] + rng.normal(0, 0.1, P[k].shape)).astype(np.float32)
    return P


def run_case(cfg, P, batch, lr, reg_lambda, steps, keep_prob=1.0, masks=None):
    """Forward intermediates + grads at step 0, params after each Adam step."""
    res = {}
    out, grads = so.loss_and_grads(cfg, P, batch, reg_lambda, keep_prob, masks)
    A = so.forward_literal(cfg, P, batch, keep_prob,
                           None if masks is None else [m.numpy() for m in masks])
    assert np.abs(A["logit"] - out["logit"].detach().numpy()).max() < 2e-5, "oracle A/B disagree"
    for k in ("user_side", "item_side", "atten_info", "user_rep", "item_rep", "att_score",
              "head_inp", "logit", "y_pred"):
        if out.get(k) is not None:
            res["fwd/" + k] = out[k].detach().numpy().astype(np.float32)
    res["fwd/loss"] = np.float32(out["loss"].item())
    res["fwd/log_loss"] = np.float32(out["log_loss"].item())
    for k, g in grads.items():
        res["grad/" + k] = g.astype(np.float32)
    params = {k: v.copy() for k, v in P.items()}
    opt = so.TFAdam(params)
    losses = []
    for s in range(steps):
        o, g = so.loss_and_grads(cfg, params, batch, reg_lambda, keep_prob, masks)
        losses.append(float(o["loss"].detach()))
        opt.step(params, g, lr)
        if s + 1 in (1, steps):
            for k, v in params.items():
                res["step%d/%s" % (s + 1, k)] = v.copy()
    res["losses"] = np.asarray(losses, dtype=np.float32)
    return res


def save(name, cfg, kw, P, batch, res, extra=None):
    blob = {"cfg": np.asarray([cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi], dtype=np.int64),
            "model_type": np.asarray(cfg.model_type)}
    for k, v in batch.items():
        blob["in/" + k] = v
    if P is not None:
        for k, v in P.items():
            blob["param/" + k] = v
    blob.update(res)
    if extra:
        blob.update(extra)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print("wrote", name, "%.1f KB" % (os.path.getsize(os.path.join(HERE, name + ".npz")) / 1e3))


def g1():
    # tiny all-features case, every model type, 3 TF-Adam steps (dense-Adam on untouched rows)
    for mt in so.MODEL_TYPES:
        w, kw = make_world("tiny")
        cfg = so.Cfg(kw["feature_size"], 4, 8, 3, 2, 3, 4, mt)
        batch = dict(zip(NAMES, w.batch(4, 0, length=2)))
        batch["length"] = np.asarray([2, 3, 1, 2], dtype=np.int32)
        P = perturbed_params(cfg, 5)
        res = run_case(cfg, P, batch, 1e-3, 5e-4, 3)
        save("g1_tiny_" + mt.lower(), cfg, kw, P, batch, res)
    # same with dropout masks (keep_prob 0.8)
    w, kw = make_world("tiny")
    cfg = so.Cfg(kw["feature_size"], 4, 8, 3, 2, 3, 4, "SCORE")
    batch = dict(zip(NAMES, w.batch(4, 0, length=2)))
    P = perturbed_params(cfg, 5)
    gen = torch.Generator().manual_seed(3)
    masks = [torch.rand((4, 200), generator=gen) < 0.8, torch.rand((4, 80), generator=gen) < 0.8]
    res = run_case(cfg, P, batch, 1e-3, 5e-4, 1, 0.8, masks)
    save("g1_tiny_score_dropout", cfg, kw, P, batch, res,
         {"in/mask0": masks[0].numpy(), "in/mask1": masks[1].numpy()})


def g3():
    # edge cases on a mid-size shape: dummy slices, length<T, duplicate ids, short batch, Fu=Fi=1
    rng = np.random.Generator(np.random.PCG64(33))
    N, D, H, T, K = 300, 8, 16, 5, 4
    for tag, Fu, Fi, B in (("f34_b3", 3, 4, 3), ("f11_b6", 1, 1, 6), ("f12_b2", 1, 2, 2)):
        cfg = so.Cfg(N, D, H, T, K, Fu, Fi, "SCORE")
        b = {"user_1hop": rng.integers(1, N, (B, T, K, Fi)), "user_2hop": rng.integers(1, N, (B, T, K, Fu)),
             "item_1hop": rng.integers(1, N, (B, T, K, Fu)), "item_2hop": rng.integers(1, N, (B, T, K, Fi)),
             "target_user": rng.integers(1, N, (B, Fu)), "target_item": rng.integers(1, N, (B, Fi)),
             "label": rng.integers(0, 2, (B,)), "length": rng.integers(1, T + 1, (B,))}
        b["user_1hop"][0, 1] = 0                      # all-dummy slice
        b["item_2hop"][0, 1] = 0                      # ... on both sides of one co-attention
        b["item_1hop"][1, :, :, :] = 0                # an entity with no history at all
        b["user_2hop"][:, :, 2:] = b["user_2hop"][:, :, :2]       # cyclic-pad duplicates
        b["user_1hop"][:, 3:] = b["user_1hop"][:, 2:3]            # tail-slice replication
        b["length"][0] = T
        b["length"][-1] = 1
        b = {k: v.astype(np.int32) for k, v in b.items()}
        P = perturbed_params(cfg, 9)
        res = run_case(cfg, P, b, 1e-3, 1e-4, 2)
        save("g3_edge_" + tag, cfg, None, P, b, res)


def g2():
    # Tmall-default shape (train_score.py:15-16,46-54,362), B=200; the 98 MB table is
    # regenerated from the seed at test time, only inputs' seed + outputs are stored.
    w, kw = make_world("tmall_default")
    cfg = so.Cfg(kw["feature_size"], 16, 32, 11, 10, 3, 4, "SCORE")
    batch = dict(zip(NAMES, w.batch(200, 0)))
    P = perturbed_params(cfg, 1111)
    out, grads = so.loss_and_grads(cfg, P, batch, 1e-4)
    touched = np.unique(np.concatenate([batch[k].ravel() for k in NAMES[:6]]))
    res = {"fwd/logit": out["logit"].detach().numpy(), "fwd/loss": np.float32(out["loss"].item()),
           "fwd/att_score": out["att_score"].detach().numpy(),
           "touched_rows": touched.astype(np.int32),
           "grad/emb_rows": grads["emb_mtx"][touched]}
    for k, g in grads.items():
        if k != "emb_mtx":
            res["grad/" + k] = g
    np.savez_compressed(os.path.join(HERE, "g2_tmall_default.npz"),
                        world=np.asarray("tmall_default"), param_seed=np.int64(1111),
                        batch_idx=np.int64(0), reg_lambda=np.float32(1e-4), **res)
    print("wrote g2", "%.1f KB" % (os.path.getsize(os.path.join(HERE, "g2_tmall_default.npz")) / 1e3))


if __name__ == "__main__":
    g1()
    g3()
    g2()
