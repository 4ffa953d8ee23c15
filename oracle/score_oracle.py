"""CPU oracle for the SCoRe forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``score_amd/`` imports it.

PARITY UNPINNED.  The reference (``/root/reference/code/score/score.py``) is a
TensorFlow-1.x graph; TensorFlow is not installed in the build container and
cannot be (no network), and the reference ships no tests, golden vectors or
checkpoints.  The arithmetic lives in the third-party
dependency ``tensorflow >= 1.4`` (``/root/reference/README.md:27``, unpinned).
This oracle therefore restates the TF graph op by op from the reference's own
call sites, and its fidelity rests on
  (a) line-by-line transcription, each function citing score.py file:line,
  (b) two independent restatements agreeing: ``forward_literal`` (Oracle A:
      NumPy fp64, materialises the [B,T,K,K,3D] tile exactly as
      score.py:147-167 does) against ``forward`` (Oracle B: collapsed form,
      torch, any dtype),
  (c) finite-difference gradient checks of Oracle B's autograd (tests/).

TF-1.x semantics assumed (reviewed, not checkable here):
  1. tf.truncated_normal_initializer defaults mean 0, stddev 1, resample
     outside 2 sigma (score.py:44).
  2. tf.layers.dense: glorot-uniform kernel, zero bias, contracts last axis.
  3. GRUCell: [r,u] = sigmoid([x,h] Wg + bg), r first; c = tanh([x, r*h] Wc + bc);
     h' = u*h + (1-u)*c; kernel rows ordered [x; h]; gate bias init 1.0.
  4. dynamic_rnn(sequence_length): for t >= len output 0, state carried.
  5. tf.layers.batch_normalization default training=False: inference affine
     with moving_mean 0 / moving_variance 1 (never updated), epsilon 1e-3:
     y = x * (gamma * rsqrt(var + eps)) + (beta - mean * gamma * rsqrt(var + eps)).
  6. tf.nn.dropout(x, keep) = x / keep * Bernoulli(keep).
  7. tf.losses.log_loss eps 1e-7, mean over batch.
  8. tf.nn.l2_loss = sum(v**2) / 2.
  9. AdamOptimizer (ApplyAdam kernel): alpha = lr*sqrt(1-b2^t)/(1-b1^t);
     m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps);
     the beta powers are fp32 variables multiplied by beta after every step.
 10. d(emb_mtx * mask) is dense, so Adam runs on every table row every step.
 11. tf.where(mask, fc3, -2**32+1) then softmax over T (score.py:179-181).
 12. softmax is max-subtracted.

Model types: SCORE and its ablations (score.py:188-369) and RRN, the slice baseline that shares the feed
tuple, table, GRUs, head, loss and optimizer (code/slice_models/slice_model.py:11-174).
"""
import math

import numpy as np
import torch

MODEL_TYPES = ("SCORE", "RIA", "RCA", "SCORE_USER", "SCORE_ITEM", "RRN")
BN_EPS = 1e-3
LOGLOSS_EPS = 1e-7
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8
PAD_SCORE = float(-2 ** 32 + 1)


class Cfg(object):
    """Constructor arguments of SCOREBASE (score.py:12-13) plus derived sizes."""

    def __init__(self, feature_size, eb_dim, hidden_size, max_time_len,
                 obj_per_time_slice, user_fnum, item_fnum, model_type="SCORE"):
        assert model_type in MODEL_TYPES
        self.N, self.D, self.H = feature_size, eb_dim, hidden_size
        self.T, self.K = max_time_len, obj_per_time_slice
        self.Fu, self.Fi = user_fnum, item_fnum
        self.Du, self.Di = user_fnum * eb_dim, item_fnum * eb_dim
        self.model_type = model_type
        # attention key width: score.py:211 (SCORE*), :285 (RCA: no atten_info)
        if model_type == "RCA":
            self.Dk = 2 * hidden_size
        elif model_type in ("RIA", "RRN"):
            self.Dk = 0
        else:
            self.Dk = 2 * hidden_size + 4 * obj_per_time_slice
        # head input width: score.py:217 / :249 / :291 / :326 / :363
        nstate = 1 if model_type in ("SCORE_USER", "SCORE_ITEM") else 2
        self.Dhead = nstate * hidden_size + self.Di + self.Du


def param_spec(cfg):
    """Trainable variables in TF creation order -> (name, shape, init, regularised).

    Order follows graph construction in score.py:188-224 (SCORE) and the
    ablations :227-369: emb_mtx (:44), co_attention denses (:155), GRU cells
    (:205-208), attention denses (:172-177), bn1/fc1-3 (:69-74).  The
    ``regularised`` flag is build_l2norm's name filter (:91-94): no 'bias', no
    'emb' in the name -- so bn1/gamma and bn1/beta ARE regularised.
    """
    c = cfg
    spec = [("emb_mtx", (c.N, c.D), "trunc_normal", False)]
    nd = 0

    def dense(i, o):
        nonlocal nd
        base = "dense" if nd == 0 else "dense_%d" % nd
        nd += 1
        spec.append((base + "/kernel", (i, o), "glorot", True))
        spec.append((base + "/bias", (o,), "zeros", False))

    if c.model_type not in ("RCA", "RRN"):
        dense(3 * c.Di, 1)          # co_attention(user_1hop, item_2hop, target_item)
        dense(3 * c.Du, 1)          # co_attention(user_2hop, item_1hop, target_user)
    for side in ("gru_user_side", "gru_item_side"):
        i = c.Di + c.Du
        if c.model_type == "RRN":   # slice_model.py:159-160: the summed 1-hop sets only
            i = c.Di if side == "gru_user_side" else c.Du
        spec.append((side + "/gru_cell/gates/kernel", (i + c.H, 2 * c.H), "glorot", True))
        spec.append((side + "/gru_cell/gates/bias", (2 * c.H,), "ones", False))
        spec.append((side + "/gru_cell/candidate/kernel", (i + c.H, c.H), "glorot", True))
        spec.append((side + "/gru_cell/candidate/bias", (c.H,), "zeros", False))
    if c.model_type not in ("RIA", "RRN"):
        dense(c.Du + c.Di, c.Dk)    # query projection, score.py:172
        dense(4 * c.Dk, 80)
        dense(80, 40)
        dense(40, 1)
    spec.append(("bn1/gamma", (c.Dhead,), "ones", True))
    spec.append(("bn1/beta", (c.Dhead,), "zeros", True))
    for name, i, o in (("fc1", c.Dhead, 200), ("fc2", 200, 80), ("fc3", 80, 1)):
        spec.append((name + "/kernel", (i, o), "glorot", True))
        spec.append((name + "/bias", (o,), "zeros", False))
    return spec


def init_params(cfg, seed=1111):
    """Initial values with the TF initialiser families (values cannot match TF's RNG)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape, init, _ in param_spec(cfg):
        if init == "trunc_normal":
            v = rng.standard_normal(shape)
            bad = np.abs(v) > 2.0
            while bad.any():
                v[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2.0
        elif init == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-lim, lim, shape)
        elif init == "ones":
            v = np.ones(shape)
        else:
            v = np.zeros(shape)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def batch_to_arrays(batch_data):
    """feed_dict conversion of score.py:102-115: nested lists (ints, float 0.0 in
    dummy slices, graph_loader.py:90-91) -> int32 ndarrays."""
    names = ("user_1hop", "user_2hop", "item_1hop", "item_2hop",
             "target_user", "target_item", "label", "length")
    return {n: np.asarray(batch_data[i]).astype(np.int32) for i, n in enumerate(names)}


# ---------------------------------------------------------------------------
# Oracle A: literal NumPy fp64 forward (materialised tile), score.py op by op
# ---------------------------------------------------------------------------
def _np_softmax(x, axis):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def _np_sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _co_attention_literal(seq1, seq2, target_t, W, b):
    """score.py:147-167 verbatim in NumPy (both seqs tiled on axis 3)."""
    B, T, K, _ = seq1.shape
    target = np.tile(target_t[:, :, None, None, :], (1, 1, K, K, 1))
    seq1_tile = np.tile(seq1[:, :, :, None, :], (1, 1, 1, K, 1))
    seq2_tile = np.tile(seq2[:, :, :, None, :], (1, 1, 1, K, 1))
    inp = np.concatenate([target, seq1_tile, seq2_tile], axis=-1)
    rel = np.maximum(inp @ W + b, 0.0)                      # [B,T,K,K,1]
    atten = _np_softmax(rel.reshape(B, T, K * K), axis=-1).reshape(B, T, K, K)
    seq1_w = atten.sum(axis=3)[..., None]
    seq2_w = atten.sum(axis=2)[..., None]
    seq1_result = (seq1 * seq1_w).sum(axis=2)
    seq2_result = (seq2 * seq2_w).sum(axis=2)
    rel = rel.reshape(B, T, K, K)
    atten_info = np.concatenate([rel.sum(axis=3), rel.sum(axis=2)], axis=2)
    return seq1_result, seq2_result, atten_info


def _gru_literal(x, length, Wg, bg, Wc, bc, H):
    """tf.nn.dynamic_rnn(GRUCell(H), sequence_length) (score.py:205-208)."""
    B, T, _ = x.shape
    h = np.zeros((B, H))
    outs = np.zeros((B, T, H))
    for t in range(T):
        gates = _np_sigmoid(np.concatenate([x[:, t], h], axis=1) @ Wg + bg)
        r, u = gates[:, :H], gates[:, H:]
        c = np.tanh(np.concatenate([x[:, t], r * h], axis=1) @ Wc + bc)
        new_h = u * h + (1.0 - u) * c
        live = (t < length)[:, None]
        outs[:, t] = np.where(live, new_h, 0.0)
        h = np.where(live, new_h, h)
    return outs, h


def _attention_literal(key, query, mask, P, names):
    """score.py:169-186; ``value`` is unused by the reference."""
    T = key.shape[1]
    q = query @ P[names[0] + "/kernel"] + P[names[0] + "/bias"]
    queries = np.tile(q[:, None, :], (1, T, 1))
    inp = np.concatenate([queries, key, queries - key, queries * key], axis=-1)
    fc1 = np.maximum(inp @ P[names[1] + "/kernel"] + P[names[1] + "/bias"], 0.0)
    fc2 = np.maximum(fc1 @ P[names[2] + "/kernel"] + P[names[2] + "/bias"], 0.0)
    fc3 = fc2 @ P[names[3] + "/kernel"] + P[names[3] + "/bias"]
    s = np.where(mask == 1.0, fc3, PAD_SCORE)
    return _np_softmax(s.reshape(-1, T), axis=-1)[..., None]


def forward_literal(cfg, params, batch, keep_prob=1.0, dropout_masks=None):
    """Oracle A.  NumPy fp64 transcription of SCOREBASE.__init__ + SCORE/ablation
    wiring (score.py:20-99, 188-369).  Returns dict of named intermediates."""
    c = cfg
    P = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    b = batch
    B = b["label"].shape[0]
    emb = P["emb_mtx"].copy()
    emb[0] = 0.0                                             # score.py:45-47
    g = lambda idx, F: emb[idx].reshape(idx.shape[:-1] + (F * c.D,))  # :51-66
    user_1hop, user_2hop = g(b["user_1hop"], c.Fi), g(b["user_2hop"], c.Fu)
    item_1hop, item_2hop = g(b["item_1hop"], c.Fu), g(b["item_2hop"], c.Fi)
    target_item, target_user = g(b["target_item"], c.Fi), g(b["target_user"], c.Fu)
    length = b["length"]
    mask = (np.arange(c.T)[None, :] < length[:, None]).astype(np.float64)[..., None]
    tu_t = np.tile(target_user[:, None, :], (1, c.T, 1))
    ti_t = np.tile(target_item[:, None, :], (1, c.T, 1))
    out = {"user_1hop": user_1hop, "user_2hop": user_2hop, "item_1hop": item_1hop,
           "item_2hop": item_2hop, "target_item": target_item, "target_user": target_user}

    if c.model_type == "RRN":                                # slice_models/slice_model.py:159-160
        user_side, item_side, atten_info = user_1hop.sum(2), item_1hop.sum(2), None
    elif c.model_type == "RCA":                              # score.py:266-269
        u1s, u2s = user_1hop.sum(2), user_2hop.sum(2)
        i1s, i2s = item_1hop.sum(2), item_2hop.sum(2)
        atten_info = None
    else:
        u1s, i2s, info_item = _co_attention_literal(
            user_1hop, item_2hop, ti_t, P["dense/kernel"], P["dense/bias"])
        u2s, i1s, info_user = _co_attention_literal(
            user_2hop, item_1hop, tu_t, P["dense_1/kernel"], P["dense_1/bias"])
        if c.model_type == "RIA":
            atten_info = info_item + info_user               # score.py:237
        else:
            atten_info = np.concatenate([info_item, info_user], axis=2)
    if c.model_type != "RRN":
        user_side = np.concatenate([u1s, u2s], axis=2)
        item_side = np.concatenate([i1s, i2s], axis=2)
    out.update(user_side=user_side, item_side=item_side, atten_info=atten_info)

    gp = lambda s, n: P[s + "/gru_cell/" + n]
    ur, uh = _gru_literal(user_side, length, gp("gru_user_side", "gates/kernel"),
                          gp("gru_user_side", "gates/bias"), gp("gru_user_side", "candidate/kernel"),
                          gp("gru_user_side", "candidate/bias"), c.H)
    ir, ih = _gru_literal(item_side, length, gp("gru_item_side", "gates/kernel"),
                          gp("gru_item_side", "gates/bias"), gp("gru_item_side", "candidate/kernel"),
                          gp("gru_item_side", "candidate/bias"), c.H)
    out.update(user_rep=ur, item_rep=ir)

    if c.model_type in ("RIA", "RRN"):                       # score.py:249 / slice_model.py:170
        inp = np.concatenate([uh, ih, target_item, target_user], axis=1)
    else:
        query = np.concatenate([target_user, target_item], axis=1)
        if c.model_type == "RCA":
            key = np.concatenate([ur, ir], axis=2)
            names = ("dense", "dense_1", "dense_2", "dense_3")
        else:
            key = np.concatenate([ur, ir, atten_info], axis=2)
            names = ("dense_2", "dense_3", "dense_4", "dense_5")
        score = _attention_literal(key, query, mask, P, names)
        uf, itf = (ur * score).sum(1), (ir * score).sum(1)
        out.update(att_score=score[..., 0], user_final=uf, item_final=itf)
        if c.model_type == "SCORE_USER":
            inp = np.concatenate([uf, target_item, target_user], axis=1)
        elif c.model_type == "SCORE_ITEM":
            inp = np.concatenate([itf, target_item, target_user], axis=1)
        else:
            inp = np.concatenate([uf, itf, target_item, target_user], axis=1)
    out["head_inp"] = inp
    # build_fc_net, score.py:68-76
    inv = P["bn1/gamma"] / np.sqrt(1.0 + BN_EPS)
    bn1 = inp * inv + P["bn1/beta"]
    fc1 = np.maximum(bn1 @ P["fc1/kernel"] + P["fc1/bias"], 0.0)
    if dropout_masks is not None:
        fc1 = fc1 * dropout_masks[0] / keep_prob
    fc2 = np.maximum(fc1 @ P["fc2/kernel"] + P["fc2/bias"], 0.0)
    if dropout_masks is not None:
        fc2 = fc2 * dropout_masks[1] / keep_prob
    logit = (fc2 @ P["fc3/kernel"] + P["fc3/bias"]).reshape(-1)
    y = _np_sigmoid(logit)
    out.update(logit=logit, y_pred=y)
    lab = b["label"].astype(np.float64)
    out["log_loss"] = float(np.mean(-lab * np.log(y + LOGLOSS_EPS)
                                    - (1 - lab) * np.log(1 - y + LOGLOSS_EPS)))
    return out


# ---------------------------------------------------------------------------
# Oracle B: collapsed form, torch (autograd gives the gradients)
# ---------------------------------------------------------------------------
# Distance of the relu pre-activations of a forward pass from the kink (tests choose seeded inputs away from it: the
# gradient of a relu network is discontinuous there, and two correct fp32 implementations whose pre-activation of ONE unit
# differs in the last bit then differ by that unit's whole gradient).  forward() resets it and reports out["relu_margin"].
# out["relu_margin_per_sample"]: the same minimum per sample [B] -- every relu unit of the graph belongs to ONE sample (nothing
# crosses the batch: batch_normalization runs in inference mode, score.py:69), so a test that wants arbitrary inputs drops the
# samples whose margin is below its threshold and compares the rest (tests/helpers.py away_from_relu_kinks).
_RELU_MARGIN = [float("inf"), None, None]      # [min |pre-activation| seen, [B, T] bool mask of the slices that count or None, per-sample minima [B] or None]


def _note_relu(pre):
    m = _RELU_MARGIN[1]
    a = pre.detach().abs()
    B = a.shape[0] if a.dim() >= 1 else 0
    if m is not None and a.dim() >= 2 and tuple(a.shape[:2]) == tuple(m.shape):
        big = torch.full_like(a, float("inf"))
        a = torch.where(m.reshape(m.shape + (1,) * (a.dim() - 2)), a, big)
    if a.numel():
        _RELU_MARGIN[0] = min(_RELU_MARGIN[0], float(a.min()))
        per = a.reshape(B, -1).min(dim=1).values.to(torch.float64)
        ps = _RELU_MARGIN[2]
        _RELU_MARGIN[2] = per if (ps is None or ps.shape != per.shape) else torch.minimum(ps, per)


def _co_attention_collapsed(seq1, seq2, tgt, W, b):
    """Exact collapsed form of score.py:147-167.

    Because both seq1 and seq2 are tiled along axis 3 (score.py:152-153),
    relateness[i,j] = r_i for every j, with
      r_i = relu(w_t.tgt + w_1.seq1_i + w_2.seq2_i + b),  W = [w_t; w_1; w_2].
    softmax over K*K of r_i then gives atten[i,j] = p_i / K with p = softmax_K(r):
      seq1_result = sum_i p_i seq1_i, seq2_result = mean_j seq2_j,
      atten_info  = [K*r_0..K*r_{K-1}, (sum_i r_i) repeated K times].
    """
    K = seq1.shape[2]
    Dx = seq1.shape[3]
    wt, w1, w2 = W[:Dx, 0], W[Dx:2 * Dx, 0], W[2 * Dx:, 0]
    c = (tgt * wt).sum(-1) + b[0]                             # [B]
    z = (seq1 * w1).sum(-1) + (seq2 * w2).sum(-1) + c[:, None, None]
    _note_relu(z)
    r = torch.relu(z)                                         # [B,T,K]
    p = torch.softmax(r, dim=-1)
    seq1_result = (seq1 * p[..., None]).sum(2)
    seq2_result = seq2.sum(2) / K
    rs = r.sum(-1, keepdim=True)
    atten_info = torch.cat([K * r, rs.expand(-1, -1, K)], dim=2)
    return seq1_result, seq2_result, atten_info


def _co_attention_tiled(seq1, seq2, tgt, W, b):
    """score.py:147-167 op for op in torch, MATERIALISING the [B,T,K,K,3Dx] tile the way the TF graph does (the
    form whose cost the reference really pays; used for the CPU timing at the Tmall-default shape and as a
    third check of the collapse).  tgt: [B, Dx] (tiled over T here, score.py:193-194)."""
    B, T, K, Dx = seq1.shape
    target = tgt[:, None, None, None, :].expand(B, T, K, K, Dx)               # :150-151 (after :193-194)
    seq1_tile = seq1[:, :, :, None, :].expand(B, T, K, K, Dx)                  # :152  tile on axis 3
    seq2_tile = seq2[:, :, :, None, :].expand(B, T, K, K, Dx)                  # :153  tile on axis 3 as well
    inp = torch.cat([target, seq1_tile, seq2_tile], dim=-1)                    # :154  [B,T,K,K,3Dx], materialised
    rel = torch.relu(inp @ W + b)                                              # :155  dense(1, relu)
    atten = torch.softmax(rel.reshape(B, T, K * K), dim=-1).reshape(B, T, K, K)  # :156-158
    seq1_w = atten.sum(3)[..., None]                                           # :159
    seq2_w = atten.sum(2)[..., None]                                           # :160
    seq1_result = (seq1 * seq1_w).sum(2)                                       # :162
    seq2_result = (seq2 * seq2_w).sum(2)                                       # :163
    rel = rel.reshape(B, T, K, K)
    atten_info = torch.cat([rel.sum(3), rel.sum(2)], dim=2)                    # :165-166
    return seq1_result, seq2_result, atten_info


def _gru(x, length, Wg, bg, Wc, bc, H):
    B, T, I = x.shape
    h = x.new_zeros((B, H))
    outs = []
    for t in range(T):
        gates = torch.sigmoid(torch.cat([x[:, t], h], 1) @ Wg + bg)
        r, u = gates[:, :H], gates[:, H:]
        c = torch.tanh(torch.cat([x[:, t], r * h], 1) @ Wc + bc)
        new_h = u * h + (1.0 - u) * c
        live = (t < length)[:, None]
        outs.append(torch.where(live, new_h, torch.zeros_like(new_h)))
        h = torch.where(live, new_h, h)
    return torch.stack(outs, 1), h


def forward(cfg, P, batch, keep_prob=1.0, dropout_masks=None, reg_lambda=0.0, tiled=False):
    """Oracle B.  ``P``: dict name -> torch tensor (any float dtype; requires_grad
    as the caller wishes).  ``batch``: dict of int64/int32 torch tensors.
    Returns dict of named intermediates incl. 'loss' (log-loss + L2).
    tiled=True runs co_attention in its literal materialised form (_co_attention_tiled)."""
    c = cfg
    coatt = _co_attention_tiled if tiled else _co_attention_collapsed
    dt = P["emb_mtx"].dtype
    emb_mask = torch.ones((c.N, 1), dtype=dt)
    emb_mask[0] = 0
    emb = P["emb_mtx"] * emb_mask                             # score.py:44-47
    # tf.nn.embedding_lookup (score.py:51-66); F.embedding is torch's row gather with a dense backward
    g = lambda idx, F: torch.nn.functional.embedding(idx.long(), emb).reshape(tuple(idx.shape[:-1]) + (F * c.D,))
    user_1hop, user_2hop = g(batch["user_1hop"], c.Fi), g(batch["user_2hop"], c.Fu)
    item_1hop, item_2hop = g(batch["item_1hop"], c.Fu), g(batch["item_2hop"], c.Fi)
    target_item, target_user = g(batch["target_item"], c.Fi), g(batch["target_user"], c.Fu)
    length = batch["length"].long()
    mask = torch.arange(c.T)[None, :] < length[:, None]       # [B,T] bool
    out = {"target_item": target_item, "target_user": target_user}
    _RELU_MARGIN[0], _RELU_MARGIN[1], _RELU_MARGIN[2] = float("inf"), mask, None     # (slices past a sample's length reach nothing)

    if c.model_type == "RRN":
        user_side, item_side, atten_info = user_1hop.sum(2), item_1hop.sum(2), None
    elif c.model_type == "RCA":
        u1s, u2s = user_1hop.sum(2), user_2hop.sum(2)
        i1s, i2s = item_1hop.sum(2), item_2hop.sum(2)
        atten_info = None
    else:
        u1s, i2s, info_item = coatt(user_1hop, item_2hop, target_item, P["dense/kernel"], P["dense/bias"])
        u2s, i1s, info_user = coatt(user_2hop, item_1hop, target_user, P["dense_1/kernel"], P["dense_1/bias"])
        atten_info = (info_item + info_user) if c.model_type == "RIA" \
            else torch.cat([info_item, info_user], 2)
    if c.model_type != "RRN":
        user_side = torch.cat([u1s, u2s], 2)
        item_side = torch.cat([i1s, i2s], 2)
    out.update(user_side=user_side, item_side=item_side, atten_info=atten_info)

    gp = lambda s, n: P[s + "/gru_cell/" + n]
    ur, uh = _gru(user_side, length, gp("gru_user_side", "gates/kernel"), gp("gru_user_side", "gates/bias"),
                  gp("gru_user_side", "candidate/kernel"), gp("gru_user_side", "candidate/bias"), c.H)
    ir, ih = _gru(item_side, length, gp("gru_item_side", "gates/kernel"), gp("gru_item_side", "gates/bias"),
                  gp("gru_item_side", "candidate/kernel"), gp("gru_item_side", "candidate/bias"), c.H)
    out.update(user_rep=ur, item_rep=ir)

    if c.model_type in ("RIA", "RRN"):
        inp = torch.cat([uh, ih, target_item, target_user], 1)
    else:
        query = torch.cat([target_user, target_item], 1)
        if c.model_type == "RCA":
            key = torch.cat([ur, ir], 2)
            n = ("dense", "dense_1", "dense_2", "dense_3")
        else:
            key = torch.cat([ur, ir, atten_info], 2)
            n = ("dense_2", "dense_3", "dense_4", "dense_5")
        q = query @ P[n[0] + "/kernel"] + P[n[0] + "/bias"]
        qs = q[:, None, :].expand(-1, c.T, -1)
        ainp = torch.cat([qs, key, qs - key, qs * key], -1)
        z1 = ainp @ P[n[1] + "/kernel"] + P[n[1] + "/bias"]
        _note_relu(z1)
        a1 = torch.relu(z1)
        z2 = a1 @ P[n[2] + "/kernel"] + P[n[2] + "/bias"]
        _note_relu(z2)
        a2 = torch.relu(z2)
        a3 = (a2 @ P[n[3] + "/kernel"] + P[n[3] + "/bias"])[..., 0]
        s = torch.where(mask, a3, torch.full_like(a3, PAD_SCORE))
        score = torch.softmax(s, dim=-1)                      # [B,T]
        uf = (ur * score[..., None]).sum(1)
        itf = (ir * score[..., None]).sum(1)
        out.update(att_score=score, user_final=uf, item_final=itf)
        if c.model_type == "SCORE_USER":
            inp = torch.cat([uf, target_item, target_user], 1)
        elif c.model_type == "SCORE_ITEM":
            inp = torch.cat([itf, target_item, target_user], 1)
        else:
            inp = torch.cat([uf, itf, target_item, target_user], 1)
    out["head_inp"] = inp
    inv = P["bn1/gamma"] * (1.0 / math.sqrt(1.0 + BN_EPS))
    bn1 = inp * inv + P["bn1/beta"]
    _RELU_MARGIN[1] = None
    zf1 = bn1 @ P["fc1/kernel"] + P["fc1/bias"]
    _note_relu(zf1)
    fc1 = torch.relu(zf1)
    if dropout_masks is not None:
        fc1 = fc1 * dropout_masks[0].to(dt) / keep_prob
    zf2 = fc1 @ P["fc2/kernel"] + P["fc2/bias"]
    _note_relu(zf2)
    fc2 = torch.relu(zf2)
    if dropout_masks is not None:
        fc2 = fc2 * dropout_masks[1].to(dt) / keep_prob
    logit = (fc2 @ P["fc3/kernel"] + P["fc3/bias"]).reshape(-1)
    y = torch.sigmoid(logit)
    lab = batch["label"].to(dt)
    log_loss = (-lab * torch.log(y + LOGLOSS_EPS)
                - (1 - lab) * torch.log(1 - y + LOGLOSS_EPS)).mean()
    l2 = sum((P[name] ** 2).sum() * 0.5 for name, _, _, reg in param_spec(c) if reg)
    out.update(logit=logit, y_pred=y, log_loss=log_loss, l2=l2,
               loss=log_loss + reg_lambda * l2, relu_margin=_RELU_MARGIN[0],
               relu_margin_per_sample=(_RELU_MARGIN[2].numpy() if _RELU_MARGIN[2] is not None else None))
    return out


def to_torch_params(params, dtype=torch.float32, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad)
            for k, v in params.items()}


def to_torch_batch(batch):
    return {k: torch.as_tensor(np.asarray(v).astype(np.int64)) for k, v in batch.items()}


def loss_and_grads(cfg, params, batch, reg_lambda, keep_prob=1.0, dropout_masks=None,
                   dtype=torch.float32, tiled=False):
    """Forward + autograd backward.  Returns (out dict, grads dict of ndarrays).
    The emb_mtx gradient is dense [N,D] with row 0 == 0 (mask, score.py:47)."""
    P = to_torch_params(params, dtype, requires_grad=True)
    out = forward(cfg, P, to_torch_batch(batch), keep_prob, dropout_masks, reg_lambda, tiled)
    out["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy()
             for k, v in P.items()}
    return out, grads


class TFAdam(object):
    """tf.train.AdamOptimizer(lr) state + ApplyAdam update (score.py:96-99), fp32.
    Every variable is updated densely every step, emb_mtx included."""

    def __init__(self, params):
        self.m = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items()}
        self.v = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items()}
        self.b1p = np.float32(ADAM_B1)
        self.b2p = np.float32(ADAM_B2)

    def alpha(self, lr):
        f = np.float32
        return f(f(lr) * np.sqrt(f(1) - self.b2p) / (f(1) - self.b1p))

    def step(self, params, grads, lr):
        f = np.float32
        a = self.alpha(lr)
        omb1, omb2 = f(1) - f(ADAM_B1), f(1) - f(ADAM_B2)
        # same fp32 operation sequence as the NumPy expressions
        #   m += (g-m)*omb1 ; v += (g*g-v)*omb2 ; p = p - (m*a)/(sqrt(v)+eps)
        # done in place through torch (multi-threaded) on the arrays' own memory
        for k in params:
            g = torch.from_numpy(np.ascontiguousarray(grads[k], dtype=np.float32))
            if not params[k].flags.writeable or not params[k].flags.c_contiguous:
                params[k] = np.array(params[k], dtype=np.float32)
            m, v, p = torch.from_numpy(self.m[k]), torch.from_numpy(self.v[k]), torch.from_numpy(params[k])
            t = g - m
            t.mul_(float(omb1))
            m.add_(t)
            torch.mul(g, g, out=t)
            t.sub_(v)
            t.mul_(float(omb2))
            v.add_(t)
            torch.mul(m, float(a), out=t)
            den = torch.sqrt(v)
            den.add_(float(f(ADAM_EPS)))
            t.div_(den)
            p.sub_(t)
        self.b1p = f(self.b1p * f(ADAM_B1))
        self.b2p = f(self.b2p * f(ADAM_B2))


class OracleModel(object):
    """CPU counterpart of SCORE(...) with the reference's train/eval signatures
    (score.py:101-133), backed by Oracle B."""

    def __init__(self, feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice,
                 user_fnum, item_fnum, model_type="SCORE", seed=1111, params=None, tiled=False):
        self.tiled = tiled      # co_attention in the literal materialised-tile form (what TF executes)
        self.cfg = Cfg(feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice,
                       user_fnum, item_fnum, model_type)
        self.params = params if params is not None else init_params(self.cfg, seed)
        self.opt = TFAdam(self.params)

    def train(self, sess, batch_data, lr, reg_lambda, keep_prob=0.8, dropout_masks=None):
        batch = batch_to_arrays(batch_data)
        B = batch["label"].shape[0]
        if dropout_masks is None and keep_prob < 1.0:
            gen = torch.Generator().manual_seed(int(self.opt.b1p * 1e9) % (2 ** 31))
            dropout_masks = [(torch.rand((B, 200), generator=gen) < keep_prob),
                             (torch.rand((B, 80), generator=gen) < keep_prob)]
        elif dropout_masks is not None:
            dropout_masks = [torch.as_tensor(np.asarray(m)) for m in dropout_masks]
        out, grads = loss_and_grads(self.cfg, self.params, batch, reg_lambda, keep_prob, dropout_masks,
                                    tiled=self.tiled)
        self.opt.step(self.params, grads, lr)
        return float(out["loss"].detach())

    def eval(self, sess, batch_data, reg_lambda):
        batch = batch_to_arrays(batch_data)
        with torch.no_grad():
            out = forward(self.cfg, to_torch_params(self.params), to_torch_batch(batch),
                          1.0, None, reg_lambda, self.tiled)
        return out["y_pred"].numpy().reshape(-1).tolist(), batch["label"].reshape(-1).tolist(), \
            float(out["loss"])
