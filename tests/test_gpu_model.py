"""GPU parity of the whole hot path through the drop-in SCORE interface (C-ABI under it)
against the committed golden vectors and the oracle (tolerances from BASELINE.json
north_star: fp32 logits within 1e-4; SURVEY 8c: grads <= 1e-4 rel)."""
import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from helpers import load_golden, batch_tuple, random_batch, NAMES

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4


def make_model(cfg, params):
    from score_amd.model import MODELS
    m = MODELS[cfg.model_type](cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi)
    m.set_params(params)
    return m


def close(a, b, rtol=1e-4, atol=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if max(np.abs(a).max(), np.abs(b).max()) < 1e-7:     # rounding noise on both sides (e.g. dense_5/bias: its gradient
        return True, 0.0                                  # is zero in exact arithmetic, the softmax is shift-invariant)
    scale = max(np.abs(b).max(), 1e-12)
    return np.abs(a - b).max() <= atol + rtol * scale, float(np.abs(a - b).max() / scale)


def adam_close(got, want, grad, max_move, atol):
    """Adam's first steps move every element by ~lr*sign(g) whatever |g| is, so where the
    gradient is at rounding-noise level the update is ill-conditioned: there only the step
    bound is asserted; everywhere else the golden value must be met to `atol`."""
    got, want, grad = np.asarray(got), np.asarray(want), np.asarray(grad)
    well = np.abs(grad) > max(1e-3 * np.abs(grad).max(), 1e-6)
    ok_well = np.abs(got - want)[well] <= atol
    ok_rest = np.abs(got - want)[~well] <= 2.2 * max_move
    return bool(ok_well.all() and ok_rest.all())


GOLD = ["g1_tiny_score", "g1_tiny_ria", "g1_tiny_rca", "g1_tiny_score_user", "g1_tiny_score_item", "g1_tiny_rrn",
        "g3_edge_f34_b3", "g3_edge_f11_b6", "g3_edge_f12_b2"]


@pytest.mark.parametrize("name", GOLD)
def test_golden_forward_backward_adam(name):
    cfg, P, b, z = load_golden(name)
    lam = 5e-4 if name.startswith("g1") else 1e-4
    m = make_model(cfg, P)
    B = b["label"].shape[0]
    # eval path: pred list, label list, loss (score.py:118-133)
    pred, label, loss = m.eval(None, batch_tuple(b), lam)
    assert isinstance(pred, list) and len(pred) == B and label == b["label"].tolist()
    assert np.abs(np.asarray(pred) - z["fwd/y_pred"]).max() < LOGIT_TOL
    assert abs(loss - float(z["fwd/loss"])) < 1e-5 * max(1.0, abs(float(z["fwd/loss"])))
    # intermediates; with skip_masked_slices the per-slice regions hold [B * A, .] rows, A = max(length): the
    # slices every sample masks (dynamic_rnn sequence_length, attention mask) are not computed at all
    T, I = cfg.T, cfg.Di + cfg.Du
    for skip in (False, True):
        m.skip_masked_slices = skip
        lay, ws = m.forward_backward(batch_tuple(b), lam, 1.0)
        torch.cuda.synchronize()
        A = min(int(b["length"].max()), T) if skip else T
        logit = ws[lay.logit:lay.logit + B].cpu().numpy()
        assert np.abs(logit - z["fwd/logit"]).max() < LOGIT_TOL
        xs = m.ws_tensor(B, "xside", (2, B * T * I))[:, :B * A * I].reshape(2, B, A, I).cpu().numpy()
        us, its = z["fwd/user_side"][:, :A], z["fwd/item_side"][:, :A]   # (RRN: the summed 1-hop sets only -- the leading columns)
        assert close(xs[0][..., :us.shape[-1]], us)[0] and close(xs[1][..., :its.shape[-1]], its)[0]
        go = m.ws_tensor(B, "gru_out", (2, B * T * cfg.H))[:, :B * A * cfg.H].reshape(2, B, A, cfg.H).cpu().numpy()
        assert close(go[0], z["fwd/user_rep"][:, :A])[0] and close(go[1], z["fwd/item_rep"][:, :A])[0]
        hi = m.ws_tensor(B, "head_inp", (B, cfg.Dhead)).cpu().numpy()
        assert close(hi, z["fwd/head_inp"])[0]
        if "fwd/att_score" in z.files:
            sc = m.ws_tensor(B, "att_score", (B, A)).cpu().numpy()
            assert close(sc, z["fwd/att_score"][:, :A], atol=1e-6)[0]
            assert not z["fwd/att_score"][:, A:].any()      # what is skipped carries exactly zero weight
        if skip:
            g_noskip = g
            g = m.get_grads()
            for k in g:      # same gradients with and without the masked slices
                ok, err = close(g[k], g_noskip[k], rtol=2e-6, atol=1e-9)
                assert ok, (k, err)
        else:
            g = m.get_grads()
    # gradients (dense emb grad with row 0 == 0; dense grads carry no L2 term: the oracle's do)
    for e in m.entries:
        want = z["grad/" + e[0]].copy()
        if e[4]:
            want = want - lam * P[e[0]]
        ok, err = close(g[e[0]], want.reshape(g[e[0]].shape), rtol=2e-4, atol=1e-7)
        assert ok, (e[0], err)
    ok, err = close(g["emb_mtx"], z["grad/emb_mtx"], rtol=2e-4, atol=1e-7)
    assert ok, err
    assert np.all(g["emb_mtx"][0] == 0)
    # three TF-Adam steps through the reference's train() signature
    m = make_model(cfg, P)
    nsteps = len(z["losses"])
    for s in range(nsteps):
        l = m.train(None, batch_tuple(b), 1e-3, lam, keep_prob=1.0)
        assert abs(l - float(z["losses"][s])) < 2e-5 * max(1.0, abs(float(z["losses"][s]))), (s, l)
        if s == 0:
            p1 = m.get_params()
            for k in P:
                assert adam_close(p1[k], z["step1/" + k], z["grad/" + k], 1e-3, 3e-6), k
    pn = m.get_params()
    for k in P:
        assert adam_close(pn[k], z["step%d/%s" % (nsteps, k)], z["grad/" + k], nsteps * 1e-3, 2e-5), k
    # dense-Adam semantics: never-touched rows and row 0 are bit-identical to the initial table
    touched = np.unique(np.concatenate([b[k].ravel() for k in NAMES[:6]]))
    never = np.setdiff1d(np.arange(cfg.N), touched)
    assert np.array_equal(pn["emb_mtx"][never], P["emb_mtx"][never])
    assert np.array_equal(pn["emb_mtx"][0], P["emb_mtx"][0])


def test_golden_dropout_masks():
    cfg, P, b, z = load_golden("g1_tiny_score_dropout")
    m = make_model(cfg, P)
    masks = [z["in/mask0"], z["in/mask1"]]
    lay, ws = m.forward_backward(batch_tuple(b), 5e-4, 0.8, masks)
    B = 4
    logit = ws[lay.logit:lay.logit + B].cpu().numpy()
    assert np.abs(logit - z["fwd/logit"]).max() < LOGIT_TOL
    g = m.get_grads()
    for e in m.entries:
        want = z["grad/" + e[0]] - (5e-4 * P[e[0]] if e[4] else 0)
        assert close(g[e[0]], want.reshape(g[e[0]].shape), rtol=2e-4, atol=1e-7)[0], e[0]
    # hashed in-kernel dropout: train() runs, loss finite, and differs from the keep_prob=1 loss
    l_drop = m.train(None, batch_tuple(b), 1e-3, 5e-4)
    assert np.isfinite(l_drop)


def test_tmall_default_shape_g2():
    # Tmall constants of train_score.py:15-16,46-54,362, B=200; table regenerated from its seed
    import os
    from score_amd.synth import make_world
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g2_tmall_default.npz"))
    w, kw = make_world(str(z["world"]))
    cfg = so.Cfg(kw["feature_size"], 16, 32, 11, 10, 3, 4, "SCORE")
    b = dict(zip(NAMES, w.batch(200, int(z["batch_idx"]))))
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                              "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    P = mg.perturbed_params(cfg, int(z["param_seed"]))
    m = make_model(cfg, P)
    lay, ws = m.forward_backward(batch_tuple(b), float(z["reg_lambda"]), 1.0)
    logit = ws[lay.logit:lay.logit + 200].cpu().numpy()
    assert np.abs(logit - z["fwd/logit"]).max() < LOGIT_TOL
    g = m.get_grads()
    rows = z["touched_rows"]
    ok, err = close(g["emb_mtx"][rows], z["grad/emb_rows"], rtol=3e-4, atol=1e-8)
    assert ok, err
    untouched = np.ones(cfg.N, dtype=bool)
    untouched[rows] = False
    assert not g["emb_mtx"][untouched].any()
    for e in m.entries:
        want = z["grad/" + e[0]] - (float(z["reg_lambda"]) * P[e[0]] if e[4] else 0)
        ok, err = close(g[e[0]], want.reshape(g[e[0]].shape), rtol=3e-4, atol=1e-8)
        assert ok, (e[0], err)


@pytest.mark.parametrize("mt", so.MODEL_TYPES)
def test_random_midsize_vs_oracle(mt):
    # seeded inputs at a size the oracle finishes in seconds; AUC parity to 4 d.p.
    from sklearn.metrics import roc_auc_score
    cfg = so.Cfg(5000, 16, 32, 7, 10, 3, 4, mt)
    rng = np.random.default_rng(11)
    P = so.init_params(cfg, 21)
    b = random_batch(rng, cfg, 96)
    b["label"] = (np.arange(96) % 2).astype(np.int32)
    m = make_model(cfg, P)
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, mt, params={k: v.copy() for k, v in P.items()})
    for _ in range(3):
        lg = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        lo = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo))
    pg, lab, _ = m.eval(None, batch_tuple(b), 1e-4)
    po, _, _ = om.eval(None, batch_tuple(b), 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL
    assert round(roc_auc_score(lab, pg), 4) == round(roc_auc_score(lab, po), 4)


@pytest.mark.parametrize("H,K,B,T", [(16, 10, 40, 5), (64, 10, 24, 6), (32, 5, 300, 4), (32, 10, 530, 3), (128, 10, 20, 7), (48, 6, 32, 5)])
def test_fused_attention_and_head_widths_vs_oracle(H, K, B, T):
    """the register-resident widths of the fused attention kernels (head_fused.hip: KQ = 36 / 84 / 44 / 52 / 148), their
    one-, two- and four-samples-per-workgroup forms (B < 512, < 1024, >= 1024 is cfg-3's test) and a shape that takes the
    separate launches (H = 48): losses, predictions and every gradient against the oracle"""
    cfg = so.Cfg(1500, 8, H, T, K, 2, 3, "SCORE")
    rng = np.random.default_rng(H + K)
    P = so.init_params(cfg, 4)
    b = random_batch(rng, cfg, B)
    m = make_model(cfg, P)
    m.debug_flags = 512          # (H = 32, B <= 512 would take the per-sample kernels instead: tests/test_gpu_persample.py)
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, "SCORE", params={k: v.copy() for k, v in P.items()})
    m.forward_backward(batch_tuple(b), 0.0, 1.0)
    g = m.get_grads()
    _, go = so.loss_and_grads(cfg, P, b, 0.0)
    for k in go:
        ok, err = close(g[k].reshape(np.asarray(go[k]).shape), go[k], rtol=2e-4, atol=2e-6)
        assert ok, (k, err)
    for _ in range(2):
        lg = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        lo = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo))
    pg, _, _ = m.eval(None, batch_tuple(b), 1e-4)
    po, _, _ = om.eval(None, batch_tuple(b), 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL


@pytest.mark.parametrize("mt", so.MODEL_TYPES)
def test_masked_slices_skipped_vs_oracle(mt):
    # every sample shorter than T (the reference's train split: 9 of 11 slices, graph_loader.py:382): the slices
    # past the longest sample are not gathered or computed (score_batch_t.active_slices) -- same losses,
    # predictions and parameters as the oracle, which computes and masks them, and as the model with the skip off
    cfg = so.Cfg(4000, 16, 32, 8, 6, 3, 4, mt)
    rng = np.random.default_rng(23)
    P = so.init_params(cfg, 4)
    b = random_batch(rng, cfg, 64)
    b["length"] = rng.integers(1, 6, 64).astype(np.int32)          # longest sample: 5 of T = 8
    b["label"] = (np.arange(64) % 2).astype(np.int32)
    m, m_all = make_model(cfg, P), make_model(cfg, P)
    m_all.skip_masked_slices = False
    assert m.device_batch(batch_tuple(b)).active_slices == 5 and m_all.device_batch(batch_tuple(b)).active_slices == 0
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, mt, params={k: v.copy() for k, v in P.items()})
    for step in range(3):
        lg = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        la = m_all.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        lo = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo)) and abs(lg - la) < 2e-6 * max(1.0, abs(la))
        if step == 0:      # one Adam step: where the gradient is above noise the two tables agree
            ga, gs = m_all.get_params()["emb_mtx"], m.get_params()["emb_mtx"]
            assert (np.abs(ga - gs) <= 3e-6).mean() > 0.999
    pg, lab, _ = m.eval(None, batch_tuple(b), 1e-4)
    po, _, _ = om.eval(None, batch_tuple(b), 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL
    # rows used only by masked slices get no gradient in the reference either: never marked, never moved
    live = np.zeros(cfg.N, bool)
    for k in NAMES[:4]:
        for i in range(64):
            live[b[k][i, :int(b["length"].max())].ravel()] = True
    for k in NAMES[4:6]:
        live[b[k].ravel()] = True
    dead_only = np.setdiff1d(np.unique(np.concatenate([b[k].ravel() for k in NAMES[:6]])), np.nonzero(live)[0])
    assert len(dead_only) > 0
    assert np.array_equal(m.get_params()["emb_mtx"][dead_only], P["emb_mtx"][dead_only])
    assert np.array_equal(m_all.get_params()["emb_mtx"][dead_only], P["emb_mtx"][dead_only])


def test_fused_and_layerwise_paths_agree():
    # the one-launch head (head_fused.hip) and attention tail against the layer-by-layer paths they replace (still
    # taken for shapes the fused kernels do not cover): same predictions, loss and gradients
    cfg = so.Cfg(3000, 16, 32, 6, 5, 3, 4, "SCORE")
    rng = np.random.default_rng(2)
    P = so.init_params(cfg, 3)
    b = random_batch(rng, cfg, 80)
    m = make_model(cfg, P)
    m.debug_flags = 512          # (both runs on the layer-by-layer pass: the per-sample kernels have neither switch)
    lay, ws = m.forward_backward(batch_tuple(b), 1e-4, 1.0)
    torch.cuda.synchronize()
    y0 = ws[lay.y_pred:lay.y_pred + 80].clone()
    l0 = float(ws[lay.loss].item())
    g0 = m.get_grads()
    # score_state_t.debug_flags bits 6 / 7: the head / the temporal attention layer by layer (rounds 1 - 3 switched these
    # through the environment, which the library reads ONCE per process: set after the first call it compared the fused
    # path with itself)
    m.debug_flags = 512 | 64 | 128
    lay, ws = m.forward_backward(batch_tuple(b), 1e-4, 1.0)
    torch.cuda.synchronize()
    y1 = ws[lay.y_pred:lay.y_pred + 80].clone()
    l1 = float(ws[lay.loss].item())
    g1 = m.get_grads()
    m.debug_flags = 0
    assert float((y0 - y1).abs().max()) < 2e-6 and abs(l0 - l1) < 2e-6
    assert not torch.equal(y0, y1) or any(not np.array_equal(g0[k], g1[k]) for k in g0)     # (the switch did switch)
    for k in g0:
        ok, err = close(g1[k], g0[k], rtol=2e-5, atol=1e-9)
        assert ok, (k, err)


@pytest.mark.parametrize("B,T", [(64, 6), (200, 11), (3, 2), (300, 20)])
def test_plan_sorts_are_interchangeable_bit_for_bit(B, T):
    """score_index_plan sorts the occurrences with csrc/sort.hip below 2 M occurrences and with the library's radix sort
    above; both are stable, so the plan -- and every sum the pull scatter builds from it -- must be the same BITS either way
    (debug_flags bit 5 forces the library, bit 8 sort.hip)"""
    cfg = so.Cfg(3000, 16, 32, T, 10, 3, 4, "SCORE")
    rng = np.random.default_rng(B)
    m = make_model(cfg, so.init_params(cfg, 3))
    b = random_batch(rng, cfg, B)
    out = {}
    for flags in (32, 256, 0):
        m.debug_flags = flags
        m.forward_backward(batch_tuple(b), 1e-4, 1.0)
        torch.cuda.synchronize()
        out[flags] = (m.dense_table_grad().clone(), m.w_g.clone())
    m.debug_flags = 0
    assert float(out[0][0].abs().max()) > 0
    for flags in (256, 0):
        assert torch.equal(out[flags][0], out[32][0]) and torch.equal(out[flags][1], out[32][1]), flags


@pytest.mark.parametrize("H,B", [(48, 96), (256, 4096)])
def test_stepwise_recurrence_hidden_sizes(H, B):
    # hidden sizes without a register-resident GRU kernel run the recurrence step by step on grouped
    # GEMMs (H = 256 is cfg-5's; B = 4096 puts its products on the bf16x3 kernel); ragged lengths
    cfg = so.Cfg(3000, 8, H, 6, 4, 3, 4, "SCORE")
    rng = np.random.default_rng(5)
    P = so.init_params(cfg, 9)
    b = random_batch(rng, cfg, B)
    b["length"] = rng.integers(1, cfg.T + 1, B).astype(np.int32)
    m = make_model(cfg, P)
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, "SCORE", params={k: v.copy() for k, v in P.items()})
    for _ in range(2):
        lg = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        lo = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo))
    pg, _, _ = m.eval(None, batch_tuple(b), 1e-4)
    po, _, _ = om.eval(None, batch_tuple(b), 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL


def test_save_restore_roundtrip(tmp_path):
    cfg, P, b, z = load_golden("g1_tiny_score")
    m = make_model(cfg, P)
    m.train(None, batch_tuple(b), 1e-3, 5e-4, keep_prob=1.0)
    path = str(tmp_path / "save_model_x" / "SCORE_4" / "ckpt")
    m.save(None, path)
    m2 = make_model(cfg, so.init_params(cfg, 99))
    m2.restore(None, path)
    l1 = m.train(None, batch_tuple(b), 1e-3, 5e-4, keep_prob=1.0)
    l2 = m2.train(None, batch_tuple(b), 1e-3, 5e-4, keep_prob=1.0)
    assert l1 == l2
    a, c = m.get_params(), m2.get_params()
    for k in a:
        assert np.array_equal(a[k], c[k]), k


def test_nested_list_feed_and_errors():
    # feed_dict contract (score.py:102-115): nested python lists with float 0.0 dummies; short last batch
    from score_amd.synth import make_world
    from score_amd.model import SCORE
    w, kw = make_world("tiny")
    m = SCORE(kw["feature_size"], 4, 8, 3, 2, 3, 4)
    bd = w.batch(4, 1, as_lists=True)
    bd[0][0][0] = np.zeros((2, 4)).tolist()        # float zeros, as user_dummy_node (graph_loader.py:90)
    p1, l1, loss1 = m.eval(None, bd, 1e-4)
    arr = tuple(np.asarray(x).astype(np.int32) for x in bd)
    p2, l2, loss2 = m.eval(None, arr, 1e-4)
    assert p1 == p2 and l1 == l2 and loss1 == loss2
    short = tuple(a[:3] for a in arr)              # odd, short batch
    p3, _, _ = m.eval(None, short, 1e-4)
    assert np.allclose(p3, p1[:3], atol=1e-6)
    with pytest.raises(ValueError):
        m.eval(None, arr[:7], 1e-4)
    with pytest.raises(ValueError):
        m.eval(None, (arr[0][:, :2],) + arr[1:], 1e-4)


def test_scatter_modes_agree_and_pull_is_deterministic():
    # sorted pull-form scatter (default) vs float atomics: same gradients; the pull form is
    # bitwise reproducible run to run (SURVEY 7 "hard parts": determinism of the scatter)
    from score_amd.synth import make_world
    from score_amd.model import SCORE
    w, kw = make_world("cfg2")
    kw.pop("batch")
    m = SCORE(**kw)
    batch = m.device_batch(w.batch(256, 3))
    m.scatter_mode = 0
    m.forward_backward(batch, 1e-4, 1.0)
    g0 = m.dense_table_grad().clone()
    w0 = m.w_g.clone()
    m.forward_backward(batch, 1e-4, 1.0)
    assert torch.equal(g0, m.dense_table_grad()) and torch.equal(w0, m.w_g)
    m.scatter_mode = 1
    m.forward_backward(batch, 1e-4, 1.0)
    g1 = m.dense_table_grad()
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-5 * scale
    assert torch.equal((g0 != 0).any(dim=1), (g1 != 0).any(dim=1))
    assert float(g0[0].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,ragged", [(64, 7, False), (50, 5, True), (33, 3, True), (256, 12, True)])
def test_streaming_recurrence_h256(B, T, ragged):
    # H = 256 (cfg-5): persistent kernel with the recurrent weights streamed from L2 in MFMA fragment order
    # (gru_stream.hip) -- against the oracle AND against the step-by-step form (debug_flags bit 0), whole and
    # partial 32-row tiles, ragged lengths
    cfg = so.Cfg(2000, 8, 256, T, 3, 3, 4, "SCORE")
    rng = np.random.default_rng(B)
    P = so.init_params(cfg, 12)
    b = random_batch(rng, cfg, B)
    b["length"] = (rng.integers(1, T + 1, B) if ragged else np.full(B, T)).astype(np.int32)
    m = make_model(cfg, P)
    ms = make_model(cfg, P)
    ms.debug_flags = 1
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, "SCORE", params={k: v.copy() for k, v in P.items()})
    pg, _, lg = m.eval(None, batch_tuple(b), 1e-4)
    ps, _, ls = ms.eval(None, batch_tuple(b), 1e-4)
    po, _, lo = om.eval(None, batch_tuple(b), 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL and abs(lg - lo) < 1e-5 * max(1.0, abs(lo))
    assert np.abs(np.asarray(pg) - np.asarray(ps)).max() < 2e-6
    m.forward_backward(batch_tuple(b), 1e-4, 1.0)
    ms.forward_backward(batch_tuple(b), 1e-4, 1.0)
    g1, g2 = m.get_grads(), ms.get_grads()
    _, go = so.loss_and_grads(cfg, P, b, 0.0)
    for k in g1:
        if np.abs(go[k]).max() <= 1e-7:      # (a gradient that is zero in exact arithmetic -- dense_5/bias under the
            assert max(np.abs(g1[k]).max(), np.abs(g2[k]).max()) < 1e-6, k      # softmax --: rounding noise on both sides)
            continue
        ok, err = close(g1[k], g2[k], rtol=2e-5, atol=1e-9)
        assert ok, (k, err)
        ok, err = close(g1[k], go[k], rtol=2e-4, atol=1e-9)
        assert ok, (k, err)
    for _ in range(2):
        l_g = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        l_o = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(l_g - l_o) < 2e-5 * max(1.0, abs(l_o))


def test_head_two_launch_form_and_x3_recurrence_against_their_references():
    # B = 544 is 34 row tiles of 16: the head's forward runs as two launches (bn1 + fc1 column slices, then fc2 / fc3 /
    # loss) -- same arithmetic per element as the one-launch form (debug_flags bit 1): bitwise equal.  H = 128 puts the
    # recurrences on the bf16x3 kernels (gru_x3.hip); debug_flags bit 2 keeps the f32-input MFMA kernels: equal to the
    # last fp32 bits, and both equal to the oracle.  B is not a multiple of 16 * 4 either: ragged tiles, ragged lengths.
    cfg = so.Cfg(3000, 8, 128, 6, 3, 3, 4, "SCORE")
    rng = np.random.default_rng(77)
    P = so.init_params(cfg, 5)
    B = 544 + 7
    b = random_batch(rng, cfg, B)
    b["length"] = rng.integers(1, cfg.T + 1, B).astype(np.int32)
    m, m1, mf = make_model(cfg, P), make_model(cfg, P), make_model(cfg, P)
    m1.debug_flags = 2
    mf.debug_flags = 4
    om = so.OracleModel(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, "SCORE", params={k: v.copy() for k, v in P.items()})
    p, _, l = m.eval(None, batch_tuple(b), 1e-4)
    p1, _, l1 = m1.eval(None, batch_tuple(b), 1e-4)
    pf, _, lf = mf.eval(None, batch_tuple(b), 1e-4)
    po, _, lo = om.eval(None, batch_tuple(b), 1e-4)
    assert p == p1 and l == l1
    assert np.abs(np.asarray(p) - np.asarray(pf)).max() < 2e-6
    assert np.abs(np.asarray(p) - np.asarray(po)).max() < LOGIT_TOL and abs(l - lo) < 1e-5 * max(1.0, abs(lo))
    for mm in (m, m1, mf):
        mm.forward_backward(batch_tuple(b), 1e-4, 0.8)
    g, g1, gf = m.get_grads(), m1.get_grads(), mf.get_grads()
    for k in g:
        assert np.array_equal(g[k], g1[k]), k
        ok, err = close(g[k], gf[k], rtol=2e-5, atol=1e-9)
        assert ok, (k, err)


def test_feed_prefetcher_and_threaded_list_walk_give_the_same_steps():
    """nested-list feed tuples (graph_loader.py:383) converted on native threads without the GIL into
    pinned staging buffers, a batch or two ahead on a worker thread (SCOREBASE.feed) -- the losses, parameters and the
    error behaviour are those of feeding the tuples one by one"""
    from score_amd.model import SCORE, DeviceBatch
    cfg = so.Cfg(3001, 16, 32, 6, 5, 3, 4, "SCORE")
    P = so.init_params(cfg, 3)
    rng = np.random.default_rng(2)
    lists = []
    for i in range(9):
        b = random_batch(rng, cfg, 40 if i % 4 else 24)          # two batch sizes: two staging rings
        t = [a.tolist() for a in batch_tuple(b)]
        t[0][1][2] = np.zeros((cfg.K, cfg.Fi)).tolist()          # float dummies as the loader writes them
        lists.append(tuple(t))
    a, b_, c = (SCORE(*[cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi], seed=7) for _ in range(3))
    for m in (a, b_, c):
        m.set_params(P)
    a.feed_threads, b_.feed_threads, c.feed_threads = 1, 6, 6
    la = [a.train(None, t, 1e-3, 1e-4) for t in lists]
    lb = [b_.train(None, t, 1e-3, 1e-4) for t in lists]
    lc = []
    for db in c.feed(iter(lists), depth=2):
        assert isinstance(db, DeviceBatch)
        lc.append(c.train(None, db, 1e-3, 1e-4))
    assert la == lb == lc
    assert torch.equal(a.w, b_.w) and torch.equal(a.w, c.w) and torch.equal(a.table, c.table)
    # a malformed tuple in the stream surfaces as the same ValueError, in the consumer, at its position
    bad = list(lists[:3])
    bad[1] = bad[1][:4] + (bad[1][4][:-1],) + bad[1][5:]
    seen = 0
    with pytest.raises(ValueError):
        for db in c.feed(iter(bad)):
            seen += 1
    assert seen == 1
    # leaving the loop early stops the worker
    for db in c.feed(iter(lists)):
        break
    import threading
    import time
    time.sleep(0.3)
    assert not [t for t in threading.enumerate() if t.name == "score-feed" and t.is_alive()]


EXTREMES = [
    # (N, D, H, T, K, Fu, Fi, B, model type): the corners of what include/score_hip.h admits
    (40, 4, 8, 1, 1, 1, 1, 1, "SCORE"),          # one sample, one slice, one neighbour, one feature, the smallest D
    (300, 8, 16, 2, 32, 2, 3, 2, "SCORE"),       # K = 32 (the largest), one target line
    (500, 256, 32, 3, 4, 1, 1, 5, "SCORE"),      # D = 256 with F = 1 (F * D / 4 = 64 slots)
    (400, 32, 64, 4, 3, 8, 8, 6, "SCORE"),       # F = 8 on both sides (F * D / 4 = 64)
    (300, 12, 24, 9, 7, 3, 2, 33, "RCA"),        # D not a power of two, odd batch, Fu > Fi
    (300, 20, 40, 5, 20, 2, 2, 17, "RIA"),       # K = 20 (the 20-wide instantiation), H = 40 (no register-resident recurrence)
    (200, 16, 32, 12, 2, 1, 5, 9, "RRN"),        # the CCMR feature split on the slice baseline
]


@pytest.mark.parametrize("N,D,H,T,K,Fu,Fi,B,mt", EXTREMES)
def test_extreme_shapes_vs_oracle(N, D, H, T, K, Fu, Fi, B, mt):
    """gradients of one pass, two TF-Adam steps and the predictions against the oracle at the corners of the shape space
    (every kernel family has its own instantiations, fall-backs and clamps there)"""
    cfg = so.Cfg(N, D, H, T, K, Fu, Fi, mt)
    rng = np.random.default_rng(N + K)
    P = so.init_params(cfg, 8)
    b = random_batch(rng, cfg, B)
    m = make_model(cfg, P)
    om = so.OracleModel(N, D, H, T, K, Fu, Fi, mt, params={k: v.copy() for k, v in P.items()})
    m.forward_backward(batch_tuple(b), 0.0, 1.0)
    g = m.get_grads()
    _, go = so.loss_and_grads(cfg, P, b, 0.0)
    for k in go:
        ok, err = close(g[k].reshape(np.asarray(go[k]).shape), go[k], rtol=3e-4, atol=2e-6)
        assert ok, (k, err)
    for _ in range(2):
        lg = m.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        lo = om.train(None, batch_tuple(b), 1e-3, 1e-4, keep_prob=1.0)
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo))
    pg, lab, _ = m.eval(None, batch_tuple(b), 1e-4)
    po, _, _ = om.eval(None, batch_tuple(b), 1e-4)
    assert lab == b["label"].tolist() and np.abs(np.asarray(pg) - np.asarray(po)).max() < LOGIT_TOL
    # the nested-list form of the same batch (threaded walk) gives the same predictions
    pl, _, _ = m.eval(None, tuple(a.tolist() for a in batch_tuple(b)), 1e-4)
    assert pl == pg
