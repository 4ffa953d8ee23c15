"""C-ABI pieces added in round 2, on the device: score_context_t (per-caller side stream + events),
score_table_init (sharding-independent initialiser), the workspace cache's eviction rule."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_table_init_is_truncated_normal_and_sharding_independent():
    from score_amd import _lib
    lib = _lib.load()
    N, D, seed = 100003, 16, 77
    p = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = torch.empty((N, D), device="cuda")
    assert lib.score_table_init(p(full), N, D, 1, 0, N, C.c_uint64(seed), s) == 0
    again = torch.empty_like(full)
    assert lib.score_table_init(p(again), N, D, 1, 0, N, C.c_uint64(seed), s) == 0
    assert torch.equal(full, again) and not bool(full[0].any())
    x = full[1:].double()
    assert float(x.abs().max()) <= 2.0
    # moments of N(0,1) truncated to [-2, 2]: mean 0, variance 1 - 4 phi(2) / (2 Phi(2) - 1) = 0.77374
    assert abs(float(x.mean())) < 3e-3 and abs(float(x.var()) - 0.773741) < 3e-3
    assert abs(float((x.abs() < 1.0).double().mean()) - 0.682689 / 0.954500) < 3e-3
    other = torch.empty_like(full)
    assert lib.score_table_init(p(other), N, D, 1, 0, N, C.c_uint64(seed + 1), s) == 0
    assert not torch.equal(other, full)
    for G in (2, 3, 8):
        rows_local = (N + G - 1) // G
        for r in range(G):
            shard = torch.full((rows_local, D), 9.0, device="cuda")
            assert lib.score_table_init(p(shard), rows_local, D, G, r, N, C.c_uint64(seed), s) == 0
            part = full[r::G]
            assert torch.equal(shard[:part.shape[0]], part) and not bool(shard[part.shape[0]:].any())
    assert lib.score_table_init(p(full), N, D, 2, 2, N, C.c_uint64(seed), s) == -1       # rank outside the stride


def test_models_own_their_context_and_run_side_by_side():
    # two models on two streams, interleaved step by step: each has its own score_context_t, so neither's side-stream
    # fork/join events are touched by the other; results equal the ones computed alone
    from oracle import score_oracle as so
    from score_amd.model import SCORE
    from helpers import random_batch, batch_tuple
    cfg = so.Cfg(2001, 8, 16, 5, 4, 3, 4, "SCORE")
    P = so.init_params(cfg, 2)
    rng = np.random.default_rng(1)
    bs = [batch_tuple(random_batch(rng, cfg, 32)) for _ in range(4)]
    args = (cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi)
    alone = SCORE(*args)
    alone.set_params(P)
    want = [alone.train(None, b, 1e-3, 1e-4, keep_prob=1.0) for b in bs]
    m1, m2 = SCORE(*args), SCORE(*args)
    assert m1._ctx.value and m2._ctx.value and m1._ctx.value != m2._ctx.value
    m1.set_params(P); m2.set_params(P)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    l1, l2 = [], []
    for b in bs:
        with torch.cuda.stream(s1):
            l1.append(m1.train_async(b, 1e-3, 1e-4, keep_prob=1.0).clone())
        with torch.cuda.stream(s2):
            l2.append(m2.train_async(b, 1e-3, 1e-4, keep_prob=1.0).clone())
    torch.cuda.synchronize()
    assert [float(x) for x in l1] == want and [float(x) for x in l2] == want
    assert torch.equal(m1.w, alone.w) and torch.equal(m2.w, alone.w) and torch.equal(m1.table, alone.table)
    ctx = m1._ctx.value
    del m1
    import gc
    gc.collect()                                  # destroys m1's context; m2's keeps working
    assert m2.train(None, bs[0], 1e-3, 1e-4, keep_prob=1.0) == alone.train(None, bs[0], 1e-3, 1e-4, keep_prob=1.0)
    assert ctx


def test_workspace_cache_evicts_one_entry_at_a_time():
    # a new (B, slot) key used to empty the whole cache, also under an index plan in flight
    from oracle import score_oracle as so
    from score_amd.model import SCORE
    from helpers import random_batch, batch_tuple
    cfg = so.Cfg(501, 4, 8, 3, 2, 3, 4, "SCORE")
    m = SCORE(cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi)
    m.max_workspaces = 4
    held = m._workspace(6, 1)                     # what a plan record keeps
    ptr = held[1].data_ptr()
    for B in range(7, 14):
        m._workspace(B)
        assert len(m._ws) <= 4
    assert (6, 1) not in m._ws and held[1].data_ptr() == ptr and held[1].numel() > 0
    m._workspace(12)                              # touching an entry makes it the most recent
    m._workspace(20)
    assert (12, 0) in m._ws and (10, 0) not in m._ws
    rng = np.random.default_rng(0)
    losses = [m.train(None, batch_tuple(random_batch(rng, cfg, B)), 1e-3, 1e-4) for B in (3, 5, 7, 9, 11, 13, 3, 5)]
    assert all(np.isfinite(l) for l in losses)


def test_stream_copy_helper_copies_exactly():
    """score_stream_copy (bench.py's measured bandwidth ceiling): a plain copy, any multiple of four floats"""
    import ctypes as C
    import torch
    from score_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (4, 1024, 4 * 100003, 1 << 24):
        src = torch.randn((n,), device="cuda")
        dst = torch.zeros((n + 8,), device="cuda")
        assert lib.score_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], src) and not bool(dst[n:].any())
    assert lib.score_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), 6, st) == -2
    assert lib.score_stream_copy(None, C.c_void_p(src.data_ptr()), 8, st) == -1


def test_step_entry_point_rejects_missing_arguments():
    """score_train_step: null structs, a struct without its events or streams -> SCORE_E_BADARG before anything is queued"""
    import ctypes as C
    from score_amd import _lib
    lib = _lib.load()
    cfg, st, bt, p = _lib.Config(), _lib.State(), _lib.Batch(), _lib.TrainStep()
    assert lib.score_train_step(None, C.byref(st), C.byref(bt), C.byref(p), None) == -1
    assert lib.score_train_step(C.byref(cfg), C.byref(st), C.byref(bt), None, None) == -1
    assert lib.score_train_step(C.byref(cfg), C.byref(st), C.byref(bt), C.byref(p), None) == -1      # (no table, no events)


def test_device_only_events_order_streams():
    """score_event_create / _record / score_stream_wait_event / _query / _synchronize (round 6: events without the system-scope
    fence, for stream-to-stream ordering) through _lib.DevEvent -- the duck type torch.cuda.Stream.wait_event accepts: a consumer
    stream that waits for the event sees what the producer stream wrote in front of the record, every time"""
    from score_amd import _lib
    dev = torch.device("cuda")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ev = _lib.DevEvent()
    x = torch.zeros(1 << 22, device=dev)
    y = torch.zeros_like(x)
    for i in range(1, 41):
        with torch.cuda.stream(a):
            x.fill_(float(i))
            ev.record(a)
        b.wait_event(ev)                       # (torch calls ev.wait(b))
        with torch.cuda.stream(b):
            y.copy_(x)
        a.wait_stream(b)                       # (the next fill must not overtake the copy)
    torch.cuda.synchronize()
    assert float(y.min()) == 40.0 and float(y.max()) == 40.0
    ev.record()
    ev.synchronize()
    assert ev.query()
    lib = _lib.load()
    assert lib.score_event_create(None) == -1 and lib.score_event_record(None, None) == -1
    assert lib.score_stream_wait_event(None, None) == -1 and lib.score_event_query(None) == -1
