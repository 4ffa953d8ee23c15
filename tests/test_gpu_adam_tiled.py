"""The time-tiled table optimizer (score_adam_touched / score_adam_catchup_ids / score_adam_catchup_rows,
include/score_hip.h) against the per-step dense ApplyAdam sweep it replaces (score.py:96-99 on emb_mtx):
the table and both Adam slots must be BIT-identical wherever they are observed -- after any number of
steps, with evals in between, through save/restore, and when switching between the two."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from helpers import batch_tuple, random_batch

pytestmark = pytest.mark.gpu


def make(cfg, window, seed=5):
    from score_amd.model import MODELS
    m = MODELS[cfg.model_type](cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, seed=seed)
    m.adam_window = window
    m.adam_tiled_min_bytes = 0          # (by default tables this small keep the per-step sweep)
    return m


def batches(cfg, n, B, seed, hot_rows=None):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        b = random_batch(rng, cfg, B)
        if hot_rows is not None and i % 3:        # most batches draw from a small hot set: the rest of the table lags
            for k in ("user_1hop", "user_2hop", "item_1hop", "item_2hop", "target_user", "target_item"):
                b[k] = np.where(b[k] > 0, 1 + b[k] % hot_rows, 0).astype(np.int32)
        out.append(batch_tuple(b))
    return out


def same_state(a, b):
    ok = torch.equal(a.table, b.table) and torch.equal(a.table_m, b.table_m) and torch.equal(a.table_v, b.table_v)
    return ok and torch.equal(a.w, b.w) and torch.equal(a.w_m, b.w_m)


@pytest.mark.parametrize("D,window", [(16, 2), (16, 5), (64, 16), (32, 7)])
def test_bitwise_equal_to_dense_sweep(D, window):
    cfg = so.Cfg(6000, D, 32, 6, 4, 2, 3, "SCORE")
    dense, tiled = make(cfg, 0), make(cfg, window)
    assert same_state(dense, tiled)
    bs = batches(cfg, 45, 24, seed=3, hot_rows=300)
    ev = batches(cfg, 2, 16, seed=9)
    for i, b in enumerate(bs):
        ld = dense.train(None, b, 1e-2, 1e-4, keep_prob=0.8)
        lt = tiled.train(None, b, 1e-2, 1e-4, keep_prob=0.8)
        assert ld == lt, (i, ld, lt)                 # the forward saw the same rows
        if i in (6, 30):                             # eval between steps reads rows the training batches may not have
            pd, _, xd = dense.eval(None, ev[0], 1e-4)
            pt, _, xt = tiled.eval(None, ev[0], 1e-4)
            assert pd == pt and xd == xt
        if i in (12, 44):
            assert tiled._adam_dirty
            assert same_state(dense, tiled), i       # (reading .table flushes)
            assert not tiled._adam_dirty
    # rows really lagged in between: the raw table differs from the flushed one right after a step
    tiled.train(None, bs[1], 1e-2, 1e-4)
    dense.train(None, bs[1], 1e-2, 1e-4)
    raw = tiled._tbl.clone()
    assert not torch.equal(raw, dense.table)
    assert same_state(dense, tiled)


def test_switching_modes_and_checkpoint(tmp_path):
    cfg = so.Cfg(3000, 16, 32, 5, 3, 2, 2, "SCORE")
    dense, tiled = make(cfg, 0), make(cfg, 4)
    bs = batches(cfg, 30, 16, seed=11, hot_rows=200)
    for i, b in enumerate(bs):
        if i == 8:
            tiled.adam_window = 0        # back to the sweep ...
        if i == 14:
            tiled.adam_window = 6        # ... and to a different window
        if i == 20:                      # through a checkpoint into a fresh object
            tiled.save(None, str(tmp_path / "ck"))
            tiled = make(cfg, 6)         # (same dropout stream; the variables come from the checkpoint)
            tiled._tbl.zero_()
            tiled.restore(None, str(tmp_path / "ck"))
        if i == 25:                      # gradients without an update in between (marks nobody consumed)
            tiled.forward_backward(b, 1e-4, 1.0)
            dense.forward_backward(b, 1e-4, 1.0)
        if i == 27:                      # somebody reads the variables between the backward and the update
            tiled.forward_backward(b, 1e-4, 0.8)
            assert tiled._adam_dirty
            tiled.get_params()           # (flushes: the rows' pending gradient marks must survive it)
            tiled.apply_adam(5e-3, 1e-4)
            dense.train(None, b, 5e-3, 1e-4)
            continue
        assert dense.train(None, b, 5e-3, 1e-4) == tiled.train(None, b, 5e-3, 1e-4), i
    assert same_state(dense, tiled)
    p_d, p_t = dense.get_params(), tiled.get_params()
    assert all(np.array_equal(p_d[k], p_t[k]) for k in p_d)


def test_fresh_rows_and_abi_errors():
    """rows the optimizer has never seen (state 0) stay bit-identical to their initial value until a batch uses them;
    the entry points reject what they cannot run"""
    from score_amd import _lib
    cfg = so.Cfg(4000, 16, 32, 4, 3, 1, 1, "SCORE")
    tiled = make(cfg, 3)
    init = tiled.table.clone()
    bs = batches(cfg, 10, 8, seed=2, hot_rows=100)
    for b in bs:
        tiled.train(None, b, 1e-2, 1e-4)
    flags = tiled.table_flags.cpu().numpy()
    t = tiled.table
    assert (flags == 0).sum() > 1000
    assert torch.equal(t[torch.from_numpy(flags == 0).to(t.device)], init[torch.from_numpy(flags == 0).to(t.device)])
    lib = _lib.load()
    _, _, T = tiled._tiled_table()
    bad = _lib.AdamTable.from_buffer_copy(T)
    bad.D = 18
    assert lib.score_adam_touched(C.byref(bad), 1, 0.1, None) != 0
    assert lib.score_adam_touched(C.byref(T), 0, 0.1, None) != 0
    assert lib.score_adam_catchup_rows(C.byref(T), 5, 3, 1, None) != 0
    assert lib.score_adam_catchup_rows(C.byref(T), 0, cfg.N + 1, 1, None) != 0
    assert lib.score_adam_catchup_ids(C.byref(T), None, 4, 1, None) != 0
    with pytest.raises(ValueError):
        m = make(cfg, 63)
        m._tiled_table()


@pytest.mark.parametrize("n_rows,D,window", [(1000, 24, 3), (777, 8, 5), (300, 256, 2), (4097, 64, 7), (65, 4, 4)])
def test_entry_points_bitwise_against_the_row_sweep(n_rows, D, window):
    """score_adam_touched / _catchup_ids / _catchup_rows driven directly through the C-ABI (row widths that leave
    lanes of a group idle, one-group-per-wave rows, a table that is not a multiple of the 64-row scan) against
    score_adam_rows on a second copy of the same state: identical bits after every flush"""
    from score_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(n_rows + D)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    blk = rnd(4, n_rows, D) * 0.1
    blk[1:3].zero_()
    ref = blk.clone()
    flags = torch.zeros(n_rows, dtype=torch.uint8, device=dev)
    flags_ref = flags.clone()
    row_step = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    ring = torch.zeros(_lib.ADAM_RING + 1, dtype=torch.float32, device=dev)
    T = _lib.AdamTable(p=P(blk[0]), m=P(blk[1]), v=P(blk[2]), g=P(blk[3]), n_rows=n_rows, D=D, row_flags=P(flags),
                       row_step=P(row_step), alpha_ring=P(ring), beta1=0.9, beta2=0.999, eps=1e-8)
    b1p = b2p = 1.0
    pending = None
    for step in range(1, 26):
        # the rows of this step's "batch" (with repeats, padding zeros and out-of-range values, as a flat batch buffer has)
        k = max(1, n_rows // (3 if step % 4 else 1))
        ids = torch.randint(0, k, (257,), device=dev, generator=g, dtype=torch.int32)
        ids[::7] = 0
        ids[3] = n_rows + 5
        ids[4] = -2
        assert lib.score_adam_catchup_ids(C.byref(T), P(ids), ids.numel(), step - 1, st) == 0
        lo, hi = n_rows * (step % window) // window, n_rows * (step % window + 1) // window
        assert lib.score_adam_catchup_rows(C.byref(T), lo, hi, step - 1, st) == 0
        # what the forward would read is what the swept copy holds
        rows = ids[(ids >= 0) & (ids < n_rows)].long().unique()
        assert torch.equal(blk[0][rows], ref[0][rows]), step
        grads = rnd(rows.numel(), D)
        for t_, f_ in ((blk, flags), (ref, flags_ref)):
            t_[3][rows] = grads
            f_[rows] = 2
        b1p *= 0.9
        b2p *= 0.999
        alpha = float(np.float32(np.float32(1e-2) * np.sqrt(np.float32(1) - np.float32(b2p)) / (np.float32(1) - np.float32(b1p))))
        assert lib.score_adam_touched(C.byref(T), step, alpha, st) == 0
        assert lib.score_adam_rows(P(ref[0]), P(ref[1]), P(ref[2]), P(ref[3]), n_rows, D, P(flags_ref), alpha, 0.9, 0.999,
                                   1e-8, None, st) == 0
        if step % 6 == 0 or step == 25:
            assert lib.score_adam_catchup_rows(C.byref(T), 0, n_rows, step, st) == 0
            assert torch.equal(blk[:3], ref[:3]), step
            assert torch.equal(flags, flags_ref)
    assert int(ring[_lib.ADAM_RING].view(torch.int32).item()) == 0


def test_captured_steps_after_tiled_ones():
    """a captured (hipGraph) step keeps the per-step sweep: switching it on after time-tiled steps applies what is
    owed first, switching it off resumes the tiled optimizer -- same bits as the sweep all the way"""
    cfg = so.Cfg(3000, 16, 32, 5, 3, 2, 2, "SCORE")
    dense, tiled = make(cfg, 0), make(cfg, 5)
    bs = batches(cfg, 16, 16, seed=21, hot_rows=150)
    for i, b in enumerate(bs):
        if i == 4:
            tiled.enable_graph(True)
        if i == 11:
            tiled.enable_graph(False)
        assert dense.train(None, b, 5e-3, 1e-4) == tiled.train(None, b, 5e-3, 1e-4), i
        if i in (3, 12):
            assert tiled._adam_dirty
        if 5 <= i <= 10:
            assert not tiled._adam_dirty
    assert same_state(dense, tiled)


def test_many_steps_wrap_the_alpha_ring():
    """more steps than the alpha ring has slots (64), with a window close to its limit: the ring is reused many times
    over and rows that are never in a batch are only ever moved by the window slice"""
    cfg = so.Cfg(2500, 16, 16, 3, 3, 1, 2, "SCORE")
    dense, tiled = make(cfg, 0), make(cfg, 31)
    bs = batches(cfg, 12, 8, seed=5, hot_rows=120)
    for i in range(200):
        b = bs[i % len(bs)]
        ld = dense.train(None, b, 1e-2, 1e-4, keep_prob=1.0)
        lt = tiled.train(None, b, 1e-2, 1e-4, keep_prob=1.0)
        assert ld == lt, i
        if i in (70, 133, 199):
            assert same_state(dense, tiled), i


# ---------------------------------------------------------------------------------------------------------------
# G6 (tests/golden/make_adam_golden.py): the optimizer the headline runs against the ORACLE's values, not against the
# HIP sweep -- 10 steps over 5 different batches, most live rows lagging, the learning rate changing once
G6 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_adam_lagging.npz")
NAMES8 = ("user_1hop", "user_2hop", "item_1hop", "item_2hop", "target_user", "target_item", "label", "length")


def _g6_errors(window):
    z = np.load(G6)
    cfg = so.Cfg(*[int(x) for x in z["cfg"]], model_type=str(z["model_type"]))
    P = so.init_params(cfg, int(z["seed"]))
    for k in list(P):
        if k != "emb_mtx":
            assert np.array_equal(P[k], z["param/" + k]), k        # (the generator's initial values)
    m = make(cfg, window)
    m.set_params(P)
    gmax = z["gmax/emb_mtx"]
    well = gmax > max(1e-3 * gmax.max(), 1e-6)                     # the adam_close rule of test_gpu_model.py
    never = gmax.max(axis=1) == 0                                  # rows no batch ever named
    errs, lr_sum = {}, 0.0
    for s, (bi, lr) in enumerate(zip(z["order"], z["lrs"])):
        bt = tuple(z["in%d/%s" % (int(bi), k)] for k in NAMES8)
        loss = m.train(None, bt, float(lr), float(z["reg_lambda"]), keep_prob=1.0)
        lr_sum += float(lr)
        assert abs(loss - float(z["losses"][s])) < 2e-5 * max(1.0, abs(float(z["losses"][s]))), (s, loss, z["losses"][s])
        if window:
            assert m._tiled_on() and m._adam_dirty
        if "step%d/emb_mtx" % (s + 1) in z.files:
            got = {"emb_mtx": m.table.cpu().numpy(), "emb_mtx/Adam": m.table_m.cpu().numpy(),
                   "emb_mtx/Adam_1": m.table_v.cpu().numpy()}
            got["emb_mtx"][0] = P["emb_mtx"][0]                    # (the masked row keeps its variable value, score.py:44-47)
            for k, g in got.items():
                w = z["step%d/%s" % (s + 1, k)]
                assert np.array_equal(g[never], w[never]), (s, k)  # untouched rows: ApplyAdam is the identity, bit for bit
                d = np.abs(g.astype(np.float64) - w)
                scale = 1.0 if k == "emb_mtx" else float(np.abs(w).max())
                errs[(s + 1, k)] = (float(d[well].max()) / scale, float(d[~well].max()) / scale, lr_sum)
    return errs, m


@pytest.mark.parametrize("window", [0, 3, 24])
def test_g6_lagging_rows_match_the_oracle(window):
    """emb_mtx and both Adam slots after steps 1, 5, 10 against oracle/score_oracle.py's TFAdam (IEEE sqrt / division, as
    TF's ApplyAdam).  window 0 = the per-step sweep, 3 / 24 = the time-tiled optimizer (24: the default; every live row
    lags until it is next read, the run is shorter than one window).  Where an element ever saw a gradient above noise:
    |d emb_mtx| <= 3e-6 after the first step and <= 5e-5 later (values are O(1); see below), the slots to 1e-5 of their
    largest value; elsewhere Adam's m / sqrt(v) turns a rounding-level gradient into a +-lr move and only the step bound
    holds."""
    errs, m = _g6_errors(window)
    rep = os.environ.get("SCORE_G6_REPORT")
    if rep:
        import json
        with open(rep, "a") as f:
            f.write(json.dumps({"window": window, "lib": os.environ.get("SCORE_HIP_LIB", "default"),
                                "errors": {"step%d/%s" % k: v[:2] for k, v in errs.items()}}) + "\n")
    for (step, k), (e_well, e_rest, lr_sum) in errs.items():
        if k == "emb_mtx":
            # after the first step: 3e-6.  Later an element's update m / (sqrt(v) + eps) depends on the RATIOS of its
            # successive gradients, so a step in which its gradient is small against its own earlier ones (a sum that
            # nearly cancels: 1e-7 absolute = 1e-3 relative) moves it by lr * 1e-3 -- measured 2.2e-5 at steps 5 and 10,
            # the same with the correctly rounded sqrt / division (tools/g6_probe.py, profiles/r03_adam_oracle_pin.md); the
            # arithmetic of the update itself is pinned to 5e-7 by the test below
            assert e_well <= (3e-6 if step == 1 else 5e-5), (step, k, e_well)
            assert e_rest <= 2.2 * lr_sum, (step, k, e_rest)
        else:
            assert e_well <= 1e-5, (step, k, e_well)


@pytest.mark.parametrize("window", [0, 3, 24])
def test_g6_optimizer_alone_on_the_oracles_gradients(window):
    """The update arithmetic by itself: the table optimizer is driven through its normal protocol (catch-up of the rows
    about to be read, the window slice, score_adam_touched / the sweep) but with the ORACLE's row gradients of every step
    written where the scatter would write them -- no forward / backward, so no gradient rounding differences.  What is
    left against so.TFAdam (IEEE sqrt and division, separately rounded multiply-adds) is score_adam1's fused
    multiply-adds and v_sqrt_f32 / v_rcp_f32: emb_mtx to 5e-7 absolute (values O(1), 4 ulp; measured 2.4e-7, and 1.2e-7
    with -DSCORE_ADAM_IEEE_DIV), the slots to 1e-6 of their largest value,
    rows nobody touched bit for bit -- after 10 steps in which up to 1,600 live rows lag."""
    z = np.load(G6)
    cfg = so.Cfg(*[int(x) for x in z["cfg"]], model_type=str(z["model_type"]))
    P = so.init_params(cfg, int(z["seed"]))
    m = make(cfg, window)
    m.set_params(P)
    cur = torch.cuda.current_stream()
    worst = {}
    for s, lr in enumerate(z["lrs"]):
        rows = torch.from_numpy(z["g%d/rows" % (s + 1)]).to(m.device)
        vals = torch.from_numpy(z["g%d/vals" % (s + 1)]).to(m.device)
        if m._tiled_on():
            m._catchup_ids([rows], True, inline_sweep=True)       # what _forward does for the batch's ids
        else:
            m._flush_adam()
        m._begin_row_grads()                                      # what forward_backward does before the scatter ...
        m.table_g[rows.long()] = vals                             # ... and the scatter itself: rows written, marked 2
        m.table_flags[rows.long()] = 2
        m.w_g.zero_()
        m.apply_adam(float(lr), 0.0)
        if window:
            assert m._adam_dirty
        if "step%d/emb_mtx" % (s + 1) in z.files:
            got = {"emb_mtx": m.table.cpu().numpy(), "emb_mtx/Adam": m.table_m.cpu().numpy(),
                   "emb_mtx/Adam_1": m.table_v.cpu().numpy()}
            got["emb_mtx"][0] = P["emb_mtx"][0]
            for k, g in got.items():
                w = z["step%d/%s" % (s + 1, k)]
                d = np.abs(g.astype(np.float64) - w)
                worst[(s + 1, k)] = float(d.max()) if k == "emb_mtx" else float(d.max() / np.abs(w).max())
    rep = os.environ.get("SCORE_G6_REPORT")
    if rep:
        import json
        with open(rep, "a") as f:
            f.write(json.dumps({"optimizer_alone": True, "window": window, "lib": os.environ.get("SCORE_HIP_LIB", "default"),
                                "errors": {"step%d/%s" % k: v for k, v in worst.items()}}) + "\n")
    for (step, k), e in worst.items():
        assert e <= (5e-7 if k == "emb_mtx" else 1e-6), (step, k, e)


def test_row_list_and_state_scan_forms_of_the_touched_update_agree():
    """score_adam_touched_rows (driven by the plan's unique-row list, model.adam_touched_list = True) against
    score_adam_touched (scan of the state bytes, the default): the same rows get the same update -- bit-identical tables --
    also when a backward pass nobody applied sits between two steps"""
    cfg = so.Cfg(6000, 32, 32, 6, 4, 2, 3, "SCORE")
    a, b = make(cfg, 5), make(cfg, 5)
    a.adam_touched_list = True
    bs = batches(cfg, 14, 24, seed=11, hot_rows=400)
    for i, bt in enumerate(bs):
        la = a.train(None, bt, 1e-2, 1e-4)
        assert a._row_list is None                       # consumed by the step
        lb = b.train(None, bt, 1e-2, 1e-4)
        assert la == lb
        if i == 6:
            a.forward_backward(bs[0], 1e-4, 1.0)         # marks + a row list that no optimizer step consumes
            assert a._row_list is not None
            b.forward_backward(bs[0], 1e-4, 1.0)
            assert b._row_list is None
    assert same_state(a, b)


@pytest.mark.parametrize("where", ["f1", "plan", "4", "2"])
def test_window_slice_placement_does_not_change_the_tables(where):
    """model.adam_sweep_at: the step's window slice beside the forward recurrence on its own stream (f1), behind the
    occurrence sort (plan) or at the last stage boundary of the backward pass (4) instead of boundary 2 -- rows nobody in the
    batch touches, any time between the batch rows' catch-up and the touched-row update: bit-identical tables and losses"""
    cfg = so.Cfg(6000, 32, 32, 6, 4, 2, 3, "SCORE")
    a, b = make(cfg, 5), make(cfg, 5)
    a.adam_sweep_at = where
    bs = batches(cfg, 16, 24, seed=13, hot_rows=400)
    for i, bt in enumerate(bs):
        la = a.train(None, bt, 1e-2, 1e-4, keep_prob=0.8)
        lb = b.train(None, bt, 1e-2, 1e-4, keep_prob=0.8)
        assert la == lb, (i, la, lb)
    if where == "f1":
        assert a._sweep_st is not None              # (the slice did run on its own stream)
    assert same_state(a, b)


def test_look_ahead_catchup_of_the_next_batch_is_bitwise_the_same():
    """apply_adam(next_batch=) / train_async(next_batch=): the next batch's rows are brought up to date THROUGH the step in
    flight on the side stream (score_adam_catchup_ids_through, behind the row scatter) instead of in front of the next
    forward.  Every (row, step) update still runs exactly once, in step order: losses and state equal the plain run's bit
    for bit -- with right hints, wrong hints (another batch comes next), no hint, an eval or a table read in between, an lr
    change, and dropout on."""
    cfg = so.Cfg(3000, 16, 16, 4, 3, 2, 3, "SCORE")
    plain, ahead = make(cfg, 5), make(cfg, 5)
    bs = batches(cfg, 16, 6, seed=9, hot_rows=150)
    dbs_p = [plain.device_batch(b) for b in bs]
    dbs_a = [ahead.device_batch(b) for b in bs]
    order = [0, 1, 2, 3, 4, 5, 0, 2, 4, 1, 3, 5, 5, 0, 1, 1, 2, 3, 4, 0, 5, 2, 3, 1, 4, 0, 1, 2, 3, 4]
    for i, bi in enumerate(order):
        lr = 1e-2 if i < 17 else 3e-3
        nxt = order[i + 1] if i + 1 < len(order) else 0
        if i % 7 == 3:
            hint = dbs_a[(nxt + 1) % len(bs)]          # a WRONG hint: some other batch comes next
        elif i % 5 == 4:
            hint = None
        else:
            hint = dbs_a[nxt]
        lp = plain.train_async(dbs_p[bi], lr, 1e-4)
        la = ahead.train_async(dbs_a[bi], lr, 1e-4, next_batch=hint)
        assert float(lp) == float(la), i
        if i > 2:
            assert ahead._tiled_on() and ahead._adam_dirty
        if i == 9:                                     # an evaluation of another batch between two steps
            assert plain.eval(None, dbs_p[4], 1e-4)[0] == ahead.eval(None, dbs_a[4], 1e-4)[0]
        if i in (13, 22):                              # somebody reads the table (flush) while a look-ahead is pending
            assert same_state(plain, ahead), i
    assert same_state(plain, ahead)
    assert ahead._ahead is None or ahead._ahead[0] is not None


def test_side_stream_finishers_and_loss_do_not_change_a_bit():
    """score_state_t.grads_done_event (the dense gradient's end-of-pass finishers on the engine's side stream, under the table's
    touched-row update) and .loss_done_event (the loss reduction there, beside the backward pass's first launches) only MOVE
    launches: with both forced on at a small shape (they engage from overlap_finishers_min_rows (b, t) rows on), losses --
    train_async's device scalar and train()'s early read-back --, gradients and optimizer state equal the run with both off,
    bit for bit; so do an evaluation in between and a forward_backward whose gradients the caller reads."""
    cfg = so.Cfg(3000, 16, 16, 4, 3, 2, 3, "SCORE")
    off, on = make(cfg, 5), make(cfg, 5)
    off.overlap_finishers_min_rows = 10 ** 9
    on.overlap_finishers_min_rows = 0
    assert on.loss_on_side and off.loss_on_side and on.dense_adam_on_side    # (the dense ApplyAdam then runs on the side stream too)
    bs = batches(cfg, 16, 5, seed=21, hot_rows=150)
    d_off, d_on = [off.device_batch(b) for b in bs], [on.device_batch(b) for b in bs]
    for i in range(14):
        bi = (3 * i) % 5
        if i % 3 == 2:          # the reference's call: the loss comes back every step (train()'s early read-back)
            assert off.train(None, d_off[bi], 1e-2, 1e-4) == on.train(None, d_on[bi], 1e-2, 1e-4), i
        else:
            assert float(off.train_async(d_off[bi], 1e-2, 1e-4)) == float(on.train_async(d_on[bi], 1e-2, 1e-4)), i
        if i >= 3:
            assert on._tiled_on() and on._ev_grads is not None and on._ev_loss is not None
            assert off._ev_grads is None and off._ev_loss is None
        if i == 6:
            assert off.eval(None, d_off[1], 1e-4)[0] == on.eval(None, d_on[1], 1e-4)[0]
        if i == 9:              # gradients read by the caller: w_g joins the finishers first
            lo, wo = off.forward_backward(d_off[2], 1e-4)
            ln, wn = on.forward_backward(d_on[2], 1e-4)
            assert torch.equal(off.w_g, on.w_g) and torch.equal(off.dense_table_grad(), on.dense_table_grad())
            assert float(wo[lo.loss]) == float(wn[ln.loss])
    assert same_state(off, on)


@pytest.mark.parametrize("mode", ["persample", "layered", "finishers", "slice_f1", "slice_plan", "slice_2", "sweep", "wide",
                                  "wide_finishers"])
def test_everything_inline_on_the_launch_stream_equals_every_overlap_mode(mode):
    """debug_flags bit 12 (4096, score_hip.h): NO second stream anywhere -- the engine's forks, the index plan (and the one
    sorted a step ahead), the window slice, the look-ahead catch-up, the dense ApplyAdam and the early loss copy all run on
    the launch stream in launch order.  Every overlap mode only MOVES launches: losses, predictions and the whole optimizer
    state equal the inline run's bit for bit, hints right or wrong, with evaluations and table reads in between."""
    H = 32 if mode in ("persample", "layered") else 16
    cfg = so.Cfg(3000, 16, H, 5, 3, 2, 3, "SCORE")
    if mode.startswith("wide"):          # a third shape: D = 64 / H = 128 (the bf16x3 products and register-resident recurrences of cfg-3)
        cfg = so.Cfg(5000, 64, 128, 5, 3, 3, 4, "SCORE")
    window = 0 if mode == "sweep" else 5
    a, b = make(cfg, window), make(cfg, window)
    for m in (a, b):
        if mode == "layered":
            m.debug_flags = 512
        if mode in ("finishers", "wide_finishers"):
            m.overlap_finishers_min_rows = 0
        if mode == "slice_f1":
            m.adam_sweep_at = "f1"
        if mode == "slice_plan":
            m.adam_sweep_at = "plan"
        if mode == "slice_2":             # (the per-sample form's placement, here on the layer-by-layer pass: "auto" gives that one "plan")
            m.adam_sweep_at = "2"
    b.debug_flags |= 4096
    ps = mode == "persample"
    assert a.persample_form(8, 5) == ps and b.persample_form(8, 5) == ps
    bs = batches(cfg, 16, 8, seed=33, hot_rows=150)
    da, db_ = [a.device_batch(x) for x in bs[:6]], [b.device_batch(x) for x in bs[:6]]
    order = [0, 1, 2, 3, 4, 5, 0, 2, 4, 1, 3, 5, 5, 0, 1, 1, 2, 3, 4, 0]
    for i, bi in enumerate(order):
        nxt = order[i + 1] if i + 1 < len(order) else 0
        hint = None if i % 5 == 4 else (nxt + 1) % 6 if i % 7 == 3 else nxt
        ha, hb = (None, None) if hint is None else (da[hint], db_[hint])
        if i % 3 == 2:
            la, lb = a.train(None, da[bi], 1e-2, 1e-4, keep_prob=0.8, next_batch=ha), b.train(None, db_[bi], 1e-2, 1e-4, keep_prob=0.8, next_batch=hb)
        else:
            la = float(a.train_async(da[bi], 1e-2, 1e-4, keep_prob=0.8, next_batch=ha))
            lb = float(b.train_async(db_[bi], 1e-2, 1e-4, keep_prob=0.8, next_batch=hb))
        assert la == lb, (mode, i, la, lb)
        cur = torch.cuda.current_stream().cuda_stream
        assert b._side is None or b._side.cuda_stream == cur
        assert b._sweep_st is None or b._sweep_st.cuda_stream == cur
        if i == 8:
            assert a.eval(None, da[4], 1e-4)[0] == b.eval(None, db_[4], 1e-4)[0]
        if i == 13:
            assert same_state(a, b), (mode, i)
    assert same_state(a, b), mode
    if mode != "sweep":
        assert a._side is not None and a._side.cuda_stream != torch.cuda.current_stream().cuda_stream
    # the switch can be thrown on a live model: the next steps of `a`, now inline, still track `b`
    a.debug_flags |= 4096
    for bi in (2, 4, 1):
        assert float(a.train_async(da[bi], 1e-2, 1e-4, keep_prob=0.8)) == float(b.train_async(db_[bi], 1e-2, 1e-4, keep_prob=0.8))
    assert same_state(a, b), mode


def test_three_hundred_overlapped_steps_equal_the_inline_run_at_a_wide_shape():
    """a soak for the layer-by-layer pass's overlap (D = 64, H = 128: cfg-3's kernels): the touched-row update beside the window
    slice and the look-ahead catch-up (publish protocol, round 5), finishers and dense ApplyAdam on side streams -- 300 steps
    with look-ahead hints against a twin with everything inline on the launch stream (debug_flags bit 12): the same bits"""
    cfg = so.Cfg(60000, 64, 128, 6, 4, 3, 4, "SCORE")
    a, b = make(cfg, 6), make(cfg, 6)
    for m in (a, b):
        m.overlap_finishers_min_rows = 0
    b.debug_flags |= 4096
    bs = batches(cfg, 16, 96, seed=91, hot_rows=5000)
    da, db_ = [a.device_batch(x) for x in bs], [b.device_batch(x) for x in bs]
    rng = np.random.default_rng(9)
    order = rng.integers(0, len(bs), 301).tolist()
    for i in range(300):
        bi, nx = order[i], order[i + 1]
        la = a.train_async(da[bi], 3e-3, 1e-4, keep_prob=0.8, next_batch=da[nx])
        lb = b.train_async(db_[bi], 3e-3, 1e-4, keep_prob=0.8, next_batch=db_[nx])
        if i % 10 == 9:
            assert float(la) == float(lb), i
        if i % 100 == 99:
            assert same_state(a, b), i
    assert same_state(a, b)
    assert a._side is not None and a._side.cuda_stream != torch.cuda.current_stream().cuda_stream and a._ev_grads is not None


@pytest.mark.parametrize("D,n_rows", [(64, 150000), (16, 400000)])
def test_touched_rows_update_beside_the_look_ahead_on_another_stream(D, n_rows):
    """score_adam_touched on one stream while score_adam_catchup_ids_through (the look-ahead catch-up of the next
    batch's rows) and score_adam_catchup_rows (the window slice) run on another -- overlapping id sets, which is how the step runs
    them since round 5 (csrc/adam_tiled.hip tiled_publish_applied: a row's count is published before its state byte).  Every
    (row, step) update must be applied exactly once whatever the interleaving: after each round the table equals a twin's on
    which the three calls ran one after the other on ONE stream, bit for bit, and every live row's count is what it must be.
    Sixty rounds on sizes whose kernels take tens of microseconds each, so that they do run side by side."""
    from score_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(D + n_rows)
    P = lambda t: C.c_void_p(t.data_ptr())
    main = torch.cuda.current_stream()
    other = torch.cuda.Stream()

    def fresh():
        blk = torch.randn(4, n_rows, D, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) * 0.1
        blk[1].abs_().mul_(0.01); blk[2].abs_().mul_(0.001)          # live moments everywhere: every row owes every step
        flags = torch.ones(n_rows, dtype=torch.uint8, device=dev)
        row_step = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        ring = torch.zeros(_lib.ADAM_RING + 1, dtype=torch.float32, device=dev)
        T = _lib.AdamTable(p=P(blk[0]), m=P(blk[1]), v=P(blk[2]), g=P(blk[3]), n_rows=n_rows, D=D, row_flags=P(flags),
                           row_step=P(row_step), alpha_ring=P(ring), beta1=0.9, beta2=0.999, eps=1e-8)
        return blk, flags, row_step, ring, T
    a, b = fresh(), fresh()
    window = 6
    ids_now = torch.randint(1, n_rows, (n_rows // 6,), device=dev, generator=gen, dtype=torch.int32)
    for m_ in (a, b):
        assert lib.score_adam_catchup_ids(C.byref(m_[4]), P(ids_now), ids_now.numel(), 0, C.c_void_p(main.cuda_stream)) == 0
    for step in range(1, 61):
        alpha = 1e-2 / (1.0 + 0.01 * step)
        # the next batch overlaps this one by about a third; both draw from a hot set now and then
        hot = (step % 3 == 0)
        ids_next = torch.randint(1, n_rows // (8 if hot else 1), (n_rows // 6,), device=dev, generator=gen, dtype=torch.int32)
        ids_next[: ids_next.numel() // 3] = ids_now[: ids_next.numel() // 3]
        rows = ids_now.long().unique()
        grads = torch.randn(rows.numel(), D, device=dev, generator=gen)
        lo, hi = n_rows * (step % window) // window, n_rows * (step % window + 1) // window
        for m_ in (a, b):
            m_[0][3][rows] = grads
            m_[1][rows] = 2                          # what the row scatter leaves: the gradient and the state-2 mark
        torch.cuda.synchronize()
        # twin b: one stream, one after the other
        sb = C.c_void_p(main.cuda_stream)
        assert lib.score_adam_catchup_rows(C.byref(b[4]), lo, hi, step - 1, sb) == 0
        assert lib.score_adam_catchup_ids_through(C.byref(b[4]), P(ids_next), ids_next.numel(), step, alpha, sb) == 0
        assert lib.score_adam_touched(C.byref(b[4]), step, alpha, sb) == 0
        # a: the touched rows on the launch stream, slice + look-ahead beside them on another
        other.wait_stream(main)
        so_ = C.c_void_p(other.cuda_stream)
        assert lib.score_adam_catchup_rows(C.byref(a[4]), lo, hi, step - 1, so_) == 0
        assert lib.score_adam_touched(C.byref(a[4]), step, alpha, sb) == 0
        assert lib.score_adam_catchup_ids_through(C.byref(a[4]), P(ids_next), ids_next.numel(), step, alpha, so_) == 0
        main.wait_stream(other)
        torch.cuda.synchronize()
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), step                   # state bytes, counts
        assert torch.equal(a[0][:3], b[0][:3]), step                                         # p, m, v
        nxt = ids_next.long().unique()
        assert bool((a[2][nxt] == step).all()) and bool((a[2][rows] == step).all()) and not bool((a[1] == 2).any())
        ids_now = ids_next
    assert int(a[3][_lib.ADAM_RING].view(torch.int32).item()) == 0
