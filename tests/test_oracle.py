"""CPU tests of the oracle itself: literal (A) vs collapsed (B), FD gradients,
golden reproduction, TF-Adam semantics.  No GPU."""
import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from helpers import load_golden, random_batch

GOLD = ["g1_tiny_score", "g1_tiny_ria", "g1_tiny_rca", "g1_tiny_score_user", "g1_tiny_score_item", "g1_tiny_rrn",
        "g3_edge_f34_b3", "g3_edge_f11_b6", "g3_edge_f12_b2"]


def _perturb(P, seed=0):
    rng = np.random.default_rng(seed)
    for k in P:
        if k != "emb_mtx":
            P[k] = (P[k] + rng.normal(0, 0.05, P[k].shape)).astype(np.float32)
    return P


@pytest.mark.parametrize("mt", so.MODEL_TYPES)
def test_literal_vs_collapsed_fp64(mt):
    # SURVEY 8c agreement rule: Oracle A (materialised [B,T,K,K,3D]) vs B <= 1e-6
    cfg = so.Cfg(97, 6, 10, 5, 7, 2, 3, mt)
    rng = np.random.default_rng(1)
    P = _perturb(so.init_params(cfg, 3))
    b = random_batch(rng, cfg, 9)
    A = so.forward_literal(cfg, P, b)
    Bo = so.forward(cfg, so.to_torch_params(P, torch.float64), so.to_torch_batch(b))
    for k in ("user_side", "item_side", "atten_info", "user_rep", "item_rep", "head_inp", "logit"):
        if A.get(k) is None:
            continue
        assert np.abs(A[k] - Bo[k].numpy()).max() < 1e-6, k
    assert abs(A["log_loss"] - float(Bo["log_loss"])) < 1e-9


def test_coattention_collapse_is_exact_property():
    # relateness[i,j] is independent of j (score.py:152-153), so w2 == 1/K exactly
    rng = np.random.default_rng(2)
    s1, s2 = rng.normal(size=(2, 3, 5, 8)), rng.normal(size=(2, 3, 5, 8))
    tg = rng.normal(size=(2, 3, 8))
    W, b = rng.normal(size=(24, 1)), rng.normal(size=(1,))
    r1, r2, info = so._co_attention_literal(s1, s2, tg, W, b)
    assert np.allclose(r2, s2.mean(axis=2), atol=1e-12)
    assert np.allclose(info[..., 5:], info[..., 5:6], atol=1e-12)


def test_fd_gradients_fp64():
    cfg = so.Cfg(40, 3, 4, 3, 3, 2, 2, "SCORE")
    rng = np.random.default_rng(5)
    P = _perturb(so.init_params(cfg, 7), 1)
    b = random_batch(rng, cfg, 3)
    lam = 1e-2
    out, grads = so.loss_and_grads(cfg, P, b, lam, dtype=torch.float64)
    tb = so.to_torch_batch(b)

    def f(Pn):
        with torch.no_grad():
            return float(so.forward(cfg, so.to_torch_params(Pn, torch.float64), tb, reg_lambda=lam)["loss"])

    eps = 1e-6
    for name in P:
        flat_idx = rng.choice(P[name].size, size=min(4, P[name].size), replace=False)
        if name == "emb_mtx":
            rows = np.unique(b["user_1hop"])[:3]
            flat_idx = [int(r) * cfg.D + 1 for r in rows]
        for fi in flat_idx:
            Pp = {k: v.astype(np.float64).copy() for k, v in P.items()}
            Pm = {k: v.astype(np.float64).copy() for k, v in P.items()}
            Pp[name].reshape(-1)[fi] += eps
            Pm[name].reshape(-1)[fi] -= eps
            fd = (f(Pp) - f(Pm)) / (2 * eps)
            an = grads[name].reshape(-1)[fi]
            if name == "emb_mtx" and fi // cfg.D == 0:
                assert an == 0.0
                continue
            assert abs(fd - an) < 1e-6 + 1e-4 * abs(fd), (name, fi, fd, an)


@pytest.mark.parametrize("name", GOLD)
def test_golden_reproduction(name):
    cfg, P, b, z = load_golden(name)
    out, grads = so.loss_and_grads(cfg, P, b, 5e-4 if name.startswith("g1") else 1e-4)
    assert np.array_equal(out["logit"].detach().numpy(), z["fwd/logit"])
    for k in grads:
        assert np.allclose(grads[k], z["grad/" + k], rtol=1e-6, atol=1e-8), k
    assert np.all(z["grad/emb_mtx"][0] == 0)


def test_tf_adam_dense_semantics():
    # fact 4 / assumption 10: a row touched once keeps moving with zero gradient;
    # a never-touched row never moves; row 0 never moves.
    cfg, P, b, z = load_golden("g1_tiny_score")
    touched = np.unique(np.concatenate([b[k].ravel() for k in b if k not in ("label", "length")]))
    never = np.setdiff1d(np.arange(cfg.N), touched)
    assert len(never) > 0
    assert np.array_equal(z["step3/emb_mtx"][never], P["emb_mtx"][never])
    assert np.array_equal(z["step3/emb_mtx"][0], P["emb_mtx"][0])
    params = {k: z["step1/" + k].copy() for k in P}
    opt = so.TFAdam(P)
    g1 = {k: z["grad/" + k] for k in P}
    p2 = {k: v.copy() for k, v in P.items()}
    opt.step(p2, g1, 1e-3)
    for k in P:
        assert np.array_equal(p2[k], params[k]), k
    zero = {k: np.zeros_like(v) for k, v in P.items()}
    before = p2["emb_mtx"].copy()
    opt.step(p2, zero, 1e-3)
    moved = np.abs(p2["emb_mtx"] - before).sum(axis=1) > 0
    t = touched[touched != 0]
    assert moved[t].all() and not moved[never].any() and not moved[0]


def test_oracle_model_interface():
    from score_amd.synth import make_world
    w, kw = make_world("tiny")
    m = so.OracleModel(kw["feature_size"], 4, 8, 3, 2, 3, 4)
    bd = w.batch(4, 0, as_lists=True)
    pred, label, loss = m.eval(None, bd, 1e-4)
    assert len(pred) == 4 and label == [1, 0, 1, 0] and np.isfinite(loss)
    l0 = m.train(None, bd, 1e-3, 1e-4, keep_prob=1.0)
    for _ in range(20):
        l1 = m.train(None, bd, 1e-3, 1e-4, keep_prob=1.0)
    assert l1 < l0


def test_tiled_torch_form_equals_collapsed_form_and_gradients():
    # third restatement: score.py:147-167 op for op in torch with the [B,T,K,K,3Dx] tile materialised (the form the
    # CPU baseline times at the Tmall-default shape) against the collapsed form, values and gradients, fp64
    cfg = so.Cfg(60, 4, 8, 3, 3, 3, 4, "SCORE")
    rng = np.random.default_rng(8)
    from helpers import random_batch
    b = random_batch(rng, cfg, 5)
    P = so.init_params(cfg, 4)
    oa, ga = so.loss_and_grads(cfg, P, b, 1e-3, dtype=torch.float64, tiled=False)
    ob, gb = so.loss_and_grads(cfg, P, b, 1e-3, dtype=torch.float64, tiled=True)
    assert abs(float(oa["loss"]) - float(ob["loss"])) < 1e-12
    assert np.abs(oa["y_pred"].detach().numpy() - ob["y_pred"].detach().numpy()).max() < 1e-12
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() < 1e-10, k


def test_relu_margin_per_sample_and_dropping_samples():
    """oracle.forward's relu_margin_per_sample: its minimum is relu_margin, a sample's value does not depend on the rest of the
    batch (nothing crosses the batch), and helpers.away_from_relu_kinks leaves a batch whose margin is at least the threshold --
    what lets the GPU parity tests run on arbitrary seeds."""
    import torch
    from helpers import away_from_relu_kinks, random_batch
    cfg = so.Cfg(300, 8, 32, 5, 4, 3, 4, "SCORE")
    P = so.init_params(cfg, 4)
    b = random_batch(np.random.default_rng(3), cfg, 64)
    with torch.no_grad():
        out = so.forward(cfg, so.to_torch_params(P), so.to_torch_batch(b))
    per = np.asarray(out["relu_margin_per_sample"])
    assert per.shape == (64,) and abs(per.min() - out["relu_margin"]) < 1e-12
    thr = 0.5 * float(np.sort(per)[4] + np.sort(per)[5])       # a threshold that drops exactly the five closest samples
    bb, _, keep = away_from_relu_kinks(cfg, P, b, thr=thr)
    assert keep.size == 59 and np.array_equal(keep, np.nonzero(per >= thr)[0])
    with torch.no_grad():
        out2 = so.forward(cfg, so.to_torch_params(P), so.to_torch_batch(bb))
    assert out2["relu_margin"] >= thr
    np.testing.assert_allclose(np.asarray(out2["relu_margin_per_sample"]), per[keep], rtol=1e-5, atol=1e-9)
