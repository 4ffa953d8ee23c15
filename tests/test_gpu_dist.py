"""The sharded (row % G) HIP path on ONE GPU: G virtual ranks run score_amd.dist.ShardedSCORE with
HipBackend in G threads of this process, exchanging through an in-process communicator with the
semantics of all_to_all_v / all_reduce.  Result must equal the single-device SCORE trained on
the concatenated batch (G ranks x B == 1 device x G*B)."""
import threading

import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from helpers import random_batch, batch_tuple, NAMES

pytestmark = pytest.mark.gpu


class ThreadGroup(object):
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm(object):
    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def exchange_counts(self, send_counts, device, extra=None):
        self.g.slots[self.rank] = (list(send_counts), extra)
        self.g.barrier.wait()
        out = [self.g.slots[p][0][self.rank] for p in range(self.world)]
        extras = [self.g.slots[p][1] for p in range(self.world)]
        self.g.barrier.wait()
        return out if extra is None else (out, extras)

    def all_to_all(self, out, inp, out_splits, in_splits):
        torch.cuda.synchronize()
        self.g.slots[self.rank] = inp.split(in_splits, 0)
        self.g.barrier.wait()
        for p, o in enumerate(out.split(out_splits, 0)):
            o.copy_(self.g.slots[p][self.rank])
        torch.cuda.synchronize()
        self.g.barrier.wait()

    def all_reduce_sum(self, t):
        torch.cuda.synchronize()
        self.g.slots[self.rank] = t.clone()
        self.g.barrier.wait()
        tot = self.g.slots[0].clone()
        for p in range(1, self.world):
            tot += self.g.slots[p]
        torch.cuda.synchronize()
        self.g.barrier.wait()
        t.copy_(tot)


def run_ranks(world, fn):
    grp = ThreadGroup(world)
    res, errs = [None] * world, []

    def body(r):
        try:
            torch.cuda.set_device(0)
            res[r] = fn(r, ThreadComm(grp, r))
        except BaseException as e:      # noqa
            errs.append(e)
            grp.barrier.abort()
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return res


@pytest.mark.parametrize("world,model_type,cfg_args,B,len_caps", [
    (2, "SCORE", (203, 4, 8, 3, 3, 3, 4), 6, None),
    (4, "SCORE", (5001, 16, 32, 5, 10, 3, 4), 16, None),
    (3, "RCA", (1000, 16, 16, 4, 5, 1, 2), 8, None),
    (2, "RIA", (777, 8, 16, 4, 4, 3, 4), 10, None),
    # every sample shorter than T, and a different longest sample per rank: each rank skips its own masked
    # slices (score_batch_t.active_slices), the single device those of the concatenated batch
    (2, "SCORE", (5001, 16, 32, 6, 5, 3, 4), 12, (3, 4)),
    (8, "SCORE", (4099, 8, 16, 5, 4, 3, 4), 6, (3, 4, 2, 4, 1, 3, 4, 4)),
])
def test_virtual_ranks_match_single_device(world, model_type, cfg_args, B, len_caps):
    from score_amd.dist import ShardedSCORE
    from score_amd.model import MODELS
    cfg = so.Cfg(*cfg_args, model_type=model_type)
    params = so.init_params(cfg, 5)
    steps = 4
    batches = [[random_batch(np.random.default_rng(100 * r + s), cfg, B) for s in range(steps)] for r in range(world)]
    if len_caps is not None:
        for r in range(world):
            for b in batches[r]:
                b["length"] = np.minimum(b["length"], len_caps[r]).astype(np.int32)

    def fn(rank, comm):
        m = ShardedSCORE(*cfg_args, comm=comm, model_type=model_type)
        m.backend.m.set_params(params)
        bts = [batch_tuple(b) for b in batches[rank]]
        losses = []
        for i, bt in enumerate(bts):            # with the next batch's index phase prefetched on a side stream
            losses.append(m.train(None, bt, 1e-3, 1e-3, keep_prob=1.0,
                                  next_batch=bts[i + 1] if i + 1 < len(bts) else None))
        pred, _, _ = m.eval(None, batch_tuple(batches[rank][0]), 1e-3)
        torch.cuda.synchronize()
        return losses, m.backend.m.table.cpu().numpy(), m.backend.m.w.cpu().numpy(), pred

    res = run_ranks(world, fn)
    ref = MODELS[model_type](*cfg_args)
    ref.set_params(params)
    for s in range(steps):
        cat = tuple(np.concatenate([batches[r][s][n] for r in range(world)]) for n in NAMES)
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            assert abs(res[r][0][s] - lref) < 2e-5 * max(1.0, abs(lref)), (s, r, res[r][0][s], lref)
    # dense replicas identical across ranks (bitwise: same all-reduced gradient, same Adam)
    for r in range(1, world):
        assert np.array_equal(res[0][2], res[r][2])
    wref = ref.w.cpu().numpy()
    close = np.abs(res[0][2] - wref) <= 3e-6
    assert close.mean() > 0.999 and np.abs(res[0][2] - wref).max() <= 2.2 * steps * 1e-3
    # shards reassemble to the single device's table
    N, D = cfg.N, cfg.D
    full = np.zeros((N, D), dtype=np.float32)
    for r in range(world):
        n_r = len(range(r, N, world))
        full[r::world] = res[r][1][:n_r]
        assert not res[r][1][n_r:].any()
    tref = ref.table.cpu().numpy()
    d = np.abs(full - tref)
    assert (d <= 3e-6).mean() > 0.999 and d.max() <= 2.2 * steps * 1e-3
    assert not full[0].any()
    # eval of rank r's batch on the sharded model == single device
    for r in range(world):
        pref, _, _ = ref.eval(None, batch_tuple(batches[r][0]), 1e-3)
        assert np.abs(np.asarray(res[r][3]) - np.asarray(pref)).max() < 1e-4


@pytest.mark.parametrize("world,cfg_args,B,pipelined", [
    (2, (5001, 16, 32, 5, 4, 3, 4), 12, True),
    (4, (5001, 16, 32, 5, 4, 3, 4), 8, False),
    (3, (3000, 32, 16, 4, 3, 1, 2), 10, True),
])
def test_tiled_shard_optimizer_bitwise_equal_to_swept(world, cfg_args, B, pipelined):
    """the time-tiled table optimizer on a row shard (the rows other ranks request are caught up before
    HipBackend.gather reads them) leaves every shard, both Adam slots and every loss bit-identical to the per-step
    sweep -- pipelined step (optimizer on the gradient-exchange stream) and plain step"""
    from score_amd.dist import ShardedSCORE
    cfg = so.Cfg(*cfg_args, model_type="SCORE")
    params = so.init_params(cfg, 9)
    steps = 14
    rng = np.random.default_rng(77)
    batches = []
    for r in range(world):
        bl = []
        for s_ in range(steps):
            b = random_batch(rng, cfg, B)
            if s_ % 3:          # most steps draw from a hot set: the rest of every shard lags
                for k in NAMES[:6]:
                    b[k] = np.where(b[k] > 0, 1 + b[k] % 400, 0).astype(np.int32)
            bl.append(b)
        batches.append(bl)

    def run(window):
        def fn(rank, comm):
            m = ShardedSCORE(*cfg_args, comm=comm, model_type="SCORE")
            sm = m.backend.m
            m.backend.auto_sweep = False       # (this test fixes the optimizer mode itself)
            sm.adam_tiled_min_bytes = 0
            sm.adam_window = window
            sm.set_params(params)
            bts = [batch_tuple(b) for b in batches[rank]]
            losses = []
            for i, bt in enumerate(bts):
                nxt = bts[i + 1] if (pipelined and i + 1 < len(bts)) else None
                losses.append(m.train(None, bt, 5e-3, 1e-3, keep_prob=1.0, next_batch=nxt))
                if i == 6:
                    pred, _, _ = m.eval(None, bts[0], 1e-3)
                    losses.append(float(np.sum(pred)))
            torch.cuda.synchronize()
            dirty = bool(sm._adam_dirty)
            return losses, sm.table.cpu().numpy(), sm.table_m.cpu().numpy(), sm.table_v.cpu().numpy(), dirty
        return run_ranks(world, fn)

    swept, tiled = run(0), run(3)
    for r in range(world):
        assert not swept[r][4] and tiled[r][4]          # (the tiled run really owed updates at the end)
        assert swept[r][0] == tiled[r][0], r
        for k in (1, 2, 3):
            assert np.array_equal(swept[r][k], tiled[r][k]), (r, k)


def test_shard_falls_back_to_the_sweep_when_most_rows_get_a_gradient():
    """eight ranks on a small table: nearly every row of a shard is requested every step, the shard's optimizer
    switches itself to the per-step sweep after four steps (HipBackend.note_requests) -- results unchanged"""
    from score_amd.dist import ShardedSCORE
    world, cfg_args, B = 8, (1600, 16, 16, 4, 4, 2, 3), 8
    cfg = so.Cfg(*cfg_args, model_type="SCORE")
    params = so.init_params(cfg, 3)
    steps = 9
    batches = [[random_batch(np.random.default_rng(1000 * r + s_), cfg, B) for s_ in range(steps)] for r in range(world)]

    def run(auto):
        def fn(rank, comm):
            m = ShardedSCORE(*cfg_args, comm=comm, model_type="SCORE")
            sm = m.backend.m
            m.backend.auto_sweep = auto
            sm.adam_tiled_min_bytes = 0
            sm.adam_window = 4 if auto else 0
            sm.set_params(params)
            bts = [batch_tuple(b) for b in batches[rank]]
            losses = [m.train(None, bt, 5e-3, 1e-3, keep_prob=1.0, next_batch=bts[i + 1] if i + 1 < steps else None)
                      for i, bt in enumerate(bts)]
            torch.cuda.synchronize()
            return losses, sm.table.cpu().numpy(), sm.table_m.cpu().numpy(), int(sm.adam_window)
        return run_ranks(world, fn)

    auto, swept = run(True), run(False)
    for r in range(world):
        assert auto[r][3] == 0, (r, auto[r][3])            # decided: sweep
        assert auto[r][0] == swept[r][0]
        assert np.array_equal(auto[r][1], swept[r][1]) and np.array_equal(auto[r][2], swept[r][2])


def test_rows_accumulate_op():
    # owner-side combine, one call per source rank in rank order: first writer stores, later ones add
    import ctypes as C
    from score_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(9)
    D, R = 24, 4000
    p = lambda t: C.c_void_p(t.data_ptr())
    out = torch.full((R, D), 7.0, device="cuda")                     # stale contents: never read
    flags = torch.zeros((R,), dtype=torch.uint8, device="cuda")
    flags[::5] = 1
    want = torch.zeros((R, D), device="cuda")
    touched = np.zeros(R, bool)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rank in range(3):
        rows = rng.choice(R, 1500, replace=False).astype(np.int32)   # unique inside a rank's list
        src = torch.from_numpy(rng.standard_normal((len(rows), D)).astype(np.float32)).cuda()
        drows = torch.from_numpy(rows).cuda()
        assert lib.score_rows_accumulate(p(drows), p(src), len(rows), D, R, p(out), p(flags), st) == 0
        want[drows.long()] += src                                    # same order of adds: bitwise equal
        touched[rows] = True
    torch.cuda.synchronize()
    t = torch.from_numpy(touched).cuda()
    assert torch.equal(out[t], want[t]) and bool((out[~t] == 7.0).all())
    wf = np.zeros(R, np.uint8); wf[::5] = 1; wf[touched] = 2
    assert np.array_equal(flags.cpu().numpy(), wf)
    assert lib.score_rows_accumulate(p(drows), p(src), 4, 6, R, p(out), p(flags), st) != 0   # D % 4


@pytest.mark.parametrize("G,D", [(2, 16), (3, 24), (8, 64), (5, 128)])
def test_rows_accumulate_multi_equals_per_source_calls(G, D):
    """score_rows_accumulate_multi -- every source rank's (unique, ascending) row list in ONE launch -- against one
    score_rows_accumulate per source in rank order: the same bits in the rows, the same state bytes; rows nobody names
    untouched; an empty source and a row named by every source included"""
    import ctypes as C
    from score_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(G * 100 + D)
    R = 5000
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lists = []
    for q in range(G):
        n = 0 if (G > 2 and q == 1) else int(rng.integers(200, 1800))
        rows = np.sort(rng.choice(R - 1, n, replace=False) + 1).astype(np.int32)
        if n:
            rows = np.unique(np.concatenate([rows, [R - 1]])).astype(np.int32)        # a row every (non-empty) source names
        lists.append(rows)
    rows_all = torch.from_numpy(np.concatenate(lists)).cuda()
    src = torch.from_numpy(rng.standard_normal((rows_all.numel(), D)).astype(np.float32)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    outs, flagss = [], []
    for mode in ("per-source", "multi"):
        out = torch.full((R, D), 7.0, device="cuda")
        flags = torch.zeros((R,), dtype=torch.uint8, device="cuda")
        flags[::3] = 1
        if mode == "per-source":
            for q in range(G):
                a, b = int(offs[q]), int(offs[q + 1])
                if b > a:
                    assert lib.score_rows_accumulate(p(rows_all[a:b]), p(src[a:b]), b - a, D, R, p(out), p(flags), st) == 0
        else:
            o = (C.c_int64 * (G + 1))(*offs.tolist())
            assert lib.score_rows_accumulate_multi(p(rows_all), p(src), o, G, D, R, p(out), p(flags), st) == 0
        torch.cuda.synchronize()
        outs.append(out); flagss.append(flags)
    assert torch.equal(outs[0], outs[1]) and torch.equal(flagss[0], flagss[1])
    named = np.zeros(R, bool); named[np.concatenate(lists)] = True
    assert bool((outs[1][torch.from_numpy(~named).cuda()] == 7.0).all()) and int((flagss[1] == 2).sum()) == int(named.sum())
    bad = (C.c_int64 * 3)(0, 5, 3)
    assert lib.score_rows_accumulate_multi(p(rows_all), p(src), bad, 2, D, R, p(outs[1]), p(flagss[1]), st) != 0     # offsets must ascend


def test_segment_sum_rows_op():
    import ctypes as C
    from score_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, D, R = 70000, 16, 5000
    rows = rng.integers(0, R, n).astype(np.int32)
    rows[:30000] = 7                      # one very hot row: exercises the long-chain path
    rows[-5:] = 0                         # local row 0 is a real row on shards > 0: summed like any other
    src = rng.standard_normal((n, D)).astype(np.float32)
    drows, dsrc = torch.from_numpy(rows).cuda(), torch.from_numpy(src).cuda()
    out = torch.full((R, D), 3.0, device="cuda")
    need = int(lib.score_segment_sum_scratch_bytes(n, D))
    scratch = torch.empty((need,), dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    flags = torch.zeros((R,), dtype=torch.uint8, device="cuda")
    flags[::3] = 1
    for it in range(2):
        rc = lib.score_segment_sum_rows(p(drows), p(dsrc), n, D, R, p(out), p(flags) if it else None, p(scratch),
                                        need, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    # every row written is marked 2; the others keep their state
    wantf = np.zeros(R, np.uint8)
    wantf[::3] = 1
    wantf[np.unique(rows)] = 2
    assert np.array_equal(flags.cpu().numpy(), wantf)
    want = np.full((R, D), 3.0, dtype=np.float64)                         # local row 0 is a real row on shards > 0: summed like any other
    drows = torch.from_numpy(rows).cuda()
    touched = np.unique(rows)
    acc = np.zeros((R, D))
    np.add.at(acc, rows, src.astype(np.float64))
    want[touched] = acc[touched]
    got = out.cpu().numpy()
    assert np.allclose(got, want, rtol=1e-5, atol=1e-3 * 1e-2 * np.abs(acc).max())


def test_cfg5_shape_eight_virtual_ranks():
    # BASELINE.json configs[4] as it is meant to run: Taobao-scale table (N = 5,042,754, D = 128) row-sharded over 8 ranks
    # (virtual: threads on the one GPU), T = 50, K = 20, H = 256, global batch 4096 = 8 x 512 -- against one device on the
    # concatenated batch.  Shards are initialised shard-locally from the seed (no rank, and no host array, ever holds
    # the 2.6 GB table); 23-bit (owner, row) plan keys.
    from score_amd.dist import ShardedSCORE
    from score_amd.model import SCORE
    from score_amd.synth import make_world
    world, steps = 8, 2
    w, kw = make_world("cfg5_taobao")
    kw.pop("batch")
    Bl = 512
    batches = [[w.batch(Bl, 700 + 10 * r + s) for s in range(steps)] for r in range(world)]

    def fn(rank, comm):
        m = ShardedSCORE(seed=31, comm=comm, **kw)
        assert m.backend.m.table.shape[0] == (kw["feature_size"] + world - 1) // world
        losses = []
        for i, bt in enumerate(batches[rank]):
            losses.append(m.train(None, bt, 1e-3, 1e-4, keep_prob=1.0,
                                  next_batch=batches[rank][i + 1] if i + 1 < steps else None))
        pred, _, _ = m.eval(None, batches[rank][0], 1e-4)
        torch.cuda.synchronize()
        rows = torch.arange(rank, kw["feature_size"], world, device="cuda")[:200000:97]       # a sample of this shard's rows
        return losses, (rows.cpu().numpy(), m.backend.m.table[(rows - rank) // world].cpu().numpy()), m.backend.m.w.cpu().numpy(), pred

    res = run_ranks(world, fn)
    ref = SCORE(seed=31, **kw)
    for s in range(steps):
        cat = tuple(np.concatenate([batches[r][s][i] for r in range(world)]) for i in range(8))
        lref = ref.train(None, cat, 1e-3, 1e-4, keep_prob=1.0)
        for r in range(world):
            assert abs(res[r][0][s] - lref) < 2e-5 * max(1.0, abs(lref)), (s, r, res[r][0][s], lref)
    for r in range(1, world):
        assert np.array_equal(res[0][2], res[r][2])
    dw = np.abs(res[0][2] - ref.w.cpu().numpy())
    assert np.median(dw) < 1e-6 and dw.max() <= 2.2 * steps * 1e-3
    for r in range(world):
        rows, vals = res[r][1]
        d = np.abs(vals - ref.table[torch.from_numpy(rows).cuda()].cpu().numpy())
        # (Adam's first steps turn a rounding-level gradient difference into a +-lr move: a few elements in a thousand
        #  differ by up to the step bound, the rest agree to rounding)
        assert (d <= 3e-6).mean() > 0.995 and d.max() <= 2.2 * steps * 1e-3
        pref, _, _ = ref.eval(None, batches[r][0], 1e-4)
        dp = np.abs(np.asarray(res[r][3]) - np.asarray(pref))
        assert np.median(dp) < 1e-4 and dp.max() < 3e-3
