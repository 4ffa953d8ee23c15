"""Feature ids outside [0, feature_size): tf.nn.embedding_lookup raises on the reference's CPU path (score.py:51-66).
Here every kernel that turns an id into an address reads such an id as the dummy row (no out-of-bounds access), the
kernels that see the ids as fed report the tensor in a sticky device word (score_state_t.id_status) and the step's
loss is NaN, so train() / eval() raise ValueError naming the tensor -- without an extra launch or read-back on the
good path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import NAMES, batch_tuple, random_batch      # noqa: E402
from oracle import score_oracle as so                      # noqa: E402

CFG = (3001, 16, 32, 6, 5, 3, 4)


def _model(mt="SCORE"):
    from score_amd.model import MODELS
    return MODELS[mt](*CFG)


@pytest.mark.parametrize("bad", [3001, -1, 2 ** 31 - 1])
@pytest.mark.parametrize("which", range(6))
def test_train_and_eval_raise_naming_the_tensor(which, bad):
    cfg = so.Cfg(*CFG, model_type="SCORE")
    m = _model()
    rng = np.random.default_rng(which)
    good = random_batch(rng, cfg, 12)
    good["length"][:] = cfg.T                       # every slice live: the bad id below is really dereferenced
    l0 = m.train(None, batch_tuple(good), 1e-3, 1e-4)
    assert np.isfinite(l0)
    b = {k: v.copy() for k, v in good.items()}
    flat = b[NAMES[which]].reshape(-1)
    flat[len(flat) // 2] = bad
    with pytest.raises(ValueError) as ei:
        m.train(None, batch_tuple(b), 1e-3, 1e-4)
    assert "batch_data[%d] (%s)" % (which, NAMES[which]) in str(ei.value)
    others = [n for i, n in enumerate(NAMES[:6]) if i != which]
    assert not any("(%s)" % n in str(ei.value) for n in others)
    with pytest.raises(ValueError):
        m.eval(None, batch_tuple(b), 1e-4)
    # the word is cleared by the raise; the model goes on (the bad id was read as the dummy row: nothing was corrupted)
    l1 = m.train(None, batch_tuple(good), 1e-3, 1e-4)
    assert np.isfinite(l1)
    p, _, le = m.eval(None, batch_tuple(good), 1e-4)
    assert np.isfinite(le) and np.isfinite(np.asarray(p)).all()
    assert torch.isfinite(m.table).all() and torch.isfinite(m.w).all()


@pytest.mark.parametrize("mt", ["RCA", "RRN", "RIA"])
def test_other_model_types_raise_too(mt):
    cfg = so.Cfg(*CFG, model_type=mt)
    m = _model(mt)
    b = random_batch(np.random.default_rng(3), cfg, 8)
    b["length"][:] = cfg.T
    b["user_1hop"][1, 2, 3, 0] = CFG[0] + 7
    b["target_item"][0, 1] = -5
    with pytest.raises(ValueError) as ei:
        m.train(None, batch_tuple(b), 1e-3, 1e-4)
    assert "(user_1hop)" in str(ei.value) and "(target_item)" in str(ei.value)


def test_bad_id_equals_dummy_row_semantics_and_async_check():
    """the offending id is read as row 0: the step's predictions equal those of the batch with 0 in its place;
    train_async / eval_async users query the status themselves"""
    cfg = so.Cfg(*CFG, model_type="SCORE")
    m = _model()
    good = random_batch(np.random.default_rng(9), cfg, 10)
    good["length"][:] = cfg.T
    bad = {k: v.copy() for k, v in good.items()}
    bad["item_1hop"][4, 1, 2, 1] = 10 ** 6
    zero = {k: v.copy() for k, v in good.items()}
    zero["item_1hop"][4, 1, 2, 1] = 0
    pz, _, _ = m.eval(None, batch_tuple(zero), 1e-4)
    yb, _, lb = m.eval_async(batch_tuple(bad), 1e-4)
    yb = yb.clone()
    assert torch.isnan(lb)
    with pytest.raises(ValueError) as ei:
        m.check_ids()
    assert "(item_1hop)" in str(ei.value)
    m.check_ids()                                   # cleared
    assert np.array_equal(yb.cpu().numpy(), np.asarray(pz, dtype=np.float32))
    # the C-ABI's synchronous query
    from score_amd import _lib
    m.eval_async(batch_tuple(bad), 1e-4)
    bits = C.c_int32(0)
    rc = m.lib.score_id_status(C.c_void_p(m._id_status.data_ptr()), C.byref(bits), 1, m._stream())
    assert rc == -4 and bits.value == 1 << 2
    assert m.lib.score_id_status(C.c_void_p(m._id_status.data_ptr()), C.byref(bits), 1, m._stream()) == 0 and bits.value == 0
    with pytest.raises(_lib.ScoreHipError):
        _lib.check(-4, "score_id_status")


def _snapshot(m):
    return dict(table=m.table.clone(), tm=m.table_m.clone(), tv=m.table_v.clone(), w=m.w.clone(), wm=m.w_m.clone(),
                wv=m.w_v.clone(), step=int(m.step), b1=np.float32(m.beta1_power), b2=np.float32(m.beta2_power))


def _same(m, snap):
    now = _snapshot(m)
    return all(torch.equal(now[k], snap[k]) for k in ("table", "tm", "tv", "w", "wm", "wv")) and \
        now["step"] == snap["step"] and now["b1"] == snap["b1"] and now["b2"] == snap["b2"]


@pytest.mark.parametrize("mode", ["sweep", "tiled", "overlap", "graph"])
def test_no_variable_is_updated_by_a_step_that_raises(mode):
    """score.py:51-66 + :101-116: tf.nn.embedding_lookup raises inside sess.run, so train_step's assigns never run -- the
    embedding table, every dense variable, both Adam slots of each and beta1_power / beta2_power are what they were before
    the call.  Here the optimizer kernels read score_state_t.id_status when they execute and apply nothing while it is set
    (score_guard_t), the host takes the step off its count again: bit for bit the state before the call, and training
    then continues exactly like a twin model that never saw the bad batch (dropout on: the per-step seed too)."""
    cfg = so.Cfg(*CFG, model_type="SCORE")
    rng = np.random.default_rng(17)
    goods = []
    for _ in range(7):
        g = random_batch(rng, cfg, 12)
        g["length"][:] = cfg.T
        goods.append(batch_tuple(g))
    bad = {k: v.copy() for k, v in zip(NAMES, goods[3])}
    bad["user_2hop"][5, 2, 1, 0] = CFG[0] + 3
    bad["target_item"][7, 2] = -9
    m, twin = _model(), _model()
    for x in (m, twin):
        if mode in ("tiled", "overlap"):
            x.adam_tiled_min_bytes = 0
        if mode == "overlap":       # the finishers, the loss reduction and the dense ApplyAdam on the side streams (what large batches take:
            x.overlap_finishers_min_rows = 0      # the layer-by-layer pass -- this shape would run as the per-sample kernels, whose
            x.debug_flags = 512                   # own side placement the sweep / tiled modes exercise)
        if mode == "graph":
            x.enable_graph(True)
    for i in range(5):                               # (graph: eager, eager, captured, replays)
        assert m.train(None, goods[i % 3], 1e-3, 1e-4) == twin.train(None, goods[i % 3], 1e-3, 1e-4)
    if mode in ("tiled", "overlap"):
        assert m._tiled_on() and m._adam_dirty
    if mode == "overlap":
        assert m._ev_grads is not None and m._fin_early      # (finishers on the engine's side stream; touched rows + dense variables in one launch behind them)
    snap = _snapshot(m)
    with pytest.raises(ValueError) as ei:
        m.train(None, batch_tuple(bad), 1e-3, 1e-4)
    assert "(user_2hop)" in str(ei.value) and "(target_item)" in str(ei.value) and "no variable was updated" in str(ei.value)
    assert _same(m, snap)
    assert int(m._id_status.sum().item()) == 0
    # ... and the run goes on as if the call had never been made
    for i in range(5, 9):
        assert m.train(None, goods[i % 7], 1e-3, 1e-4) == twin.train(None, goods[i % 7], 1e-3, 1e-4), i
    assert _same(m, _snapshot(twin))


@pytest.mark.parametrize("mode", ["sweep", "tiled", "overlap"])
def test_async_steps_queued_behind_a_bad_batch_are_not_applied_either(mode):
    """train_async: the host has queued more steps by the time anybody looks.  The word is sticky, so the device applies
    none of them and counts them; check_ids() takes exactly that many off the host's step count."""
    cfg = so.Cfg(*CFG, model_type="SCORE")
    rng = np.random.default_rng(23)
    goods = []
    for _ in range(4):
        g = random_batch(rng, cfg, 10)
        g["length"][:] = cfg.T
        goods.append(batch_tuple(g))
    bad = {k: v.copy() for k, v in zip(NAMES, goods[1])}
    bad["item_2hop"][0, 0, 0, 0] = 2 ** 30
    m, twin = _model(), _model()
    for x in (m, twin):
        if mode in ("tiled", "overlap"):
            x.adam_tiled_min_bytes = 0
        if mode == "overlap":
            x.overlap_finishers_min_rows = 0
            x.debug_flags = 512
    for i in range(3):
        m.train_async(goods[i], 1e-3, 1e-4)
        twin.train_async(goods[i], 1e-3, 1e-4)
    snap = _snapshot(m)
    losses = [m.train_async(batch_tuple(bad), 1e-3, 1e-4).clone()]
    losses += [m.train_async(goods[i], 1e-3, 1e-4).clone() for i in (3, 0)]
    assert all(bool(torch.isnan(x)) for x in losses)           # every queued step reports it
    with pytest.raises(ValueError) as ei:
        m.check_ids()
    assert "(item_2hop)" in str(ei.value) and "2 queued behind it" in str(ei.value)
    assert _same(m, snap)
    for i in (3, 0, 1):
        assert float(m.train_async(goods[i], 1e-3, 1e-4)) == float(twin.train_async(goods[i], 1e-3, 1e-4))
    assert _same(m, _snapshot(twin))


def test_look_ahead_hint_naming_a_bad_batch_changes_nothing():
    """apply_adam(next_batch=) catches the NEXT batch's rows up one step early: a hinted batch with an id outside the table
    must neither be dereferenced out of bounds there (the id scan ignores such values) nor change what the raise leaves behind"""
    cfg = so.Cfg(*CFG, model_type="SCORE")
    rng = np.random.default_rng(41)
    m, twin = _model(), _model()
    for x in (m, twin):
        x.adam_tiled_min_bytes = 0
    goods = []
    for _ in range(4):
        g = random_batch(rng, cfg, 10)
        g["length"][:] = cfg.T
        goods.append(g)
    bad = {k: v.copy() for k, v in goods[2].items()}
    bad["item_1hop"][3, 2, 1, 0] = 2 ** 31 - 1
    bad["user_2hop"][0, 0, 0, 1] = -7
    dg = [m.device_batch(batch_tuple(g)) for g in goods]
    dt = [twin.device_batch(batch_tuple(g)) for g in goods]
    db = m.device_batch(batch_tuple(bad))
    for i in range(3):
        nxt = db if i == 2 else dg[i + 1]
        assert float(m.train_async(dg[i], 1e-3, 1e-4, next_batch=nxt)) == float(twin.train_async(dt[i], 1e-3, 1e-4, next_batch=dt[(i + 1) % 4]))
    snap = _snapshot(m)
    with pytest.raises(ValueError) as ei:
        m.train(None, db, 1e-3, 1e-4)
    assert "(item_1hop)" in str(ei.value) and "(user_2hop)" in str(ei.value)
    assert _same(m, snap)
    for i in (3, 0, 1):
        assert m.train(None, dg[i], 1e-3, 1e-4) == twin.train(None, dt[i], 1e-3, 1e-4), i
    assert _same(m, _snapshot(twin))


def test_a_flush_with_the_word_set_raises_instead_of_returning_a_stale_table():
    cfg = so.Cfg(*CFG, model_type="SCORE")
    m = _model()
    m.adam_tiled_min_bytes = 0
    g = random_batch(np.random.default_rng(2), cfg, 10)
    g["length"][:] = cfg.T
    for _ in range(3):
        m.train_async(batch_tuple(g), 1e-3, 1e-4)
    bad = {k: v.copy() for k, v in g.items()}
    bad["user_1hop"][1, 1, 1, 1] = CFG[0]
    m.train_async(batch_tuple(bad), 1e-3, 1e-4)
    with pytest.raises(ValueError):
        m.table                                               # (save / get_params / tests read it through the same property)
    assert torch.isfinite(m.table).all() and int(m.step) == 3


def test_masked_slices_are_not_dereferenced():
    """ids of slices every sample masks (t >= max length) are never turned into addresses: garbage there is harmless
    and -- documented in include/score_hip.h -- not reported"""
    cfg = so.Cfg(*CFG, model_type="SCORE")
    m = _model()
    b = random_batch(np.random.default_rng(5), cfg, 8)
    b["length"][:] = 3
    ref = m.eval(None, batch_tuple(b), 1e-4)[0]
    b2 = {k: v.copy() for k, v in b.items()}
    b2["user_2hop"][:, 4:, :, :] = 2 ** 30
    assert m.eval(None, batch_tuple(b2), 1e-4)[0] == ref
    assert np.isfinite(m.train(None, batch_tuple(b2), 1e-3, 1e-4))


def test_tiled_optimizer_and_sharded_plan_report():
    """the time-tiled table optimizer's id scan ignores such ids; the sharded path's index plan reports them (its
    forward only sees unique positions) and routes them to the dummy row"""
    from score_amd.dist import ShardedSCORE
    cfg = so.Cfg(*CFG, model_type="SCORE")
    m = _model()
    m.adam_tiled_min_bytes = 0
    good = random_batch(np.random.default_rng(11), cfg, 10)
    good["length"][:] = cfg.T
    bad = {k: v.copy() for k, v in good.items()}
    bad["user_1hop"][0, 0, 0, 0] = CFG[0]
    for _ in range(3):
        m.train(None, batch_tuple(good), 1e-3, 1e-4)
    with pytest.raises(ValueError):
        m.train(None, batch_tuple(bad), 1e-3, 1e-4)
    assert np.isfinite(m.train(None, batch_tuple(good), 1e-3, 1e-4)) and torch.isfinite(m.table).all()

    class OneRank(object):           # a one-rank communicator without a process group
        rank, world = 0, 1

        def exchange_counts(self, send, device, extra=None):
            return (list(send), [extra]) if extra is not None else list(send)

        def all_to_all(self, out, inp, out_splits, in_splits):
            out.copy_(inp)

        def all_reduce_sum(self, t):
            pass
    s = ShardedSCORE(*CFG, comm=OneRank())
    assert np.isfinite(s.train(None, batch_tuple(good), 1e-3, 1e-4))
    with pytest.raises(ValueError) as ei:
        s.train(None, batch_tuple(bad), 1e-3, 1e-4)
    assert "(user_1hop)" in str(ei.value) and "rank 0" in str(ei.value) and "before its step started" in str(ei.value)
    assert np.isfinite(s.train(None, batch_tuple(good), 1e-3, 1e-4))
    # rejected on the host before anything ran: a twin that never saw the bad batch holds the same bits
    t = ShardedSCORE(*CFG, comm=OneRank())
    for _ in range(2):
        t.train(None, batch_tuple(good), 1e-3, 1e-4)
    assert torch.equal(s.backend.m.table, t.backend.m.table) and torch.equal(s.backend.m.w, t.backend.m.w)
    with pytest.raises(ValueError):
        s.eval(None, batch_tuple(bad), 1e-4)
    pe, _, le = s.eval(None, batch_tuple(good), 1e-4)
    assert np.isfinite(le) and np.isfinite(np.asarray(pe)).all()
