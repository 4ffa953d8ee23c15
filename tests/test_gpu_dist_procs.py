"""The sharded HIP path with REAL processes: two ranks, each its own process with its own ShardedSCORE + HipBackend,
sharing the one GPU of the box and exchanging through gloo (device tensors staged through host memory by
TorchDistComm's gloo branch).  Same check as the virtual-rank test: G ranks x B must equal one device x G*B.
Covers what threads cannot: per-process stream / event state, the pipelined step's prefetch across real
collectives, process start-up order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CFG_ARGS = (3001, 16, 32, 6, 5, 3, 4)
STEPS, B = 4, 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(rank):
    sys.path.insert(0, HERE)
    from oracle import score_oracle as so
    from helpers import random_batch
    cfg = so.Cfg(*CFG_ARGS, model_type="SCORE")
    out = [random_batch(np.random.default_rng(500 * rank + s), cfg, B) for s in range(STEPS)]
    for b in out:                                   # every sample shorter than T, a different longest one per rank
        b["length"] = np.minimum(b["length"], 4 + rank).astype(np.int32)
    return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from helpers import batch_tuple
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = so.Cfg(*CFG_ARGS, model_type="SCORE")
    params = so.init_params(cfg, 5)
    model = ShardedSCORE(*CFG_ARGS, comm=TorchDistComm())
    model.backend.m.set_params(params)
    bts = [batch_tuple(b) for b in _batches(rank)]
    losses = []
    for i, bt in enumerate(bts):
        losses.append(model.train(None, bt, 1e-3, 1e-3, keep_prob=1.0, next_batch=bts[i + 1] if i + 1 < len(bts) else None))
    pred, _, _ = model.eval(None, bts[0], 1e-3)
    torch.cuda.synchronize()
    # per-rank checkpoint (score.py:135-142): a fresh model restored from it continues bit for bit
    ck = os.path.join(out_dir, "save_model_x", "SCORE_24", "ckpt")
    model.save(None, ck)
    twin = ShardedSCORE(*CFG_ARGS, comm=model.comm, seed=99)
    twin.restore(None, ck)
    l_a = model.train(None, bts[1], 1e-3, 1e-3, keep_prob=1.0)
    l_b = twin.train(None, bts[1], 1e-3, 1e-3, keep_prob=1.0)
    assert l_a == l_b and torch.equal(model.backend.m.table, twin.backend.m.table) and torch.equal(model.backend.m.w, twin.backend.m.w)
    # batch sizes changing under the look-ahead pipeline -- 5 sizes x 3 plan slots = more (B, slot) workspaces
    # than the cache holds (max_workspaces lowered to force evictions while plans are in flight)
    model.backend.m.max_workspaces = 4
    vb = _var_batches(rank)
    vlosses = []
    for i, bt in enumerate(vb):
        vlosses.append(model.train(None, bt, 1e-3, 1e-3, keep_prob=1.0, next_batch=vb[i + 1] if i + 1 < len(vb) else None))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.asarray(losses), pred=np.asarray(pred),
             vlosses=np.asarray(vlosses),
             table=model.backend.m.table.cpu().numpy(), w=model.backend.m.w.cpu().numpy())
    dist.destroy_process_group()


VAR_B = (24, 20, 16, 12, 8, 24, 16, 8, 20, 12, 24, 20)


def _var_batches(rank):
    sys.path.insert(0, HERE)
    from oracle import score_oracle as so
    from helpers import random_batch, batch_tuple
    cfg = so.Cfg(*CFG_ARGS, model_type="SCORE")
    return [batch_tuple(random_batch(np.random.default_rng(9000 + 100 * rank + s), cfg, b)) for s, b in enumerate(VAR_B)]


def test_two_processes_share_the_gpu(tmp_path):
    from oracle import score_oracle as so
    from score_amd.model import SCORE
    from helpers import NAMES, batch_tuple
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    cfg = so.Cfg(*CFG_ARGS, model_type="SCORE")
    ref = SCORE(*CFG_ARGS)
    ref.set_params(so.init_params(cfg, 5))
    per_rank = [_batches(r) for r in range(world)]
    for s in range(STEPS):
        cat = tuple(np.concatenate([per_rank[r][s][n] for r in range(world)]) for n in NAMES)
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            assert abs(z[r]["losses"][s] - lref) < 2e-5 * max(1.0, abs(lref)), (s, r, z[r]["losses"][s], lref)
    for r in range(world):
        pref, _, _ = ref.eval(None, batch_tuple(per_rank[r][0]), 1e-3)
        assert np.abs(z[r]["pred"] - np.asarray(pref)).max() < 1e-4
    # (the workers then ran one more step on batch 1 for the checkpoint check: replay it here)
    cat = tuple(np.concatenate([per_rank[r][1][n] for r in range(world)]) for n in NAMES)
    ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
    # ... and the run of changing batch sizes under look-ahead
    vb = [_var_batches(r) for r in range(world)]
    for s in range(len(VAR_B)):
        cat = tuple(np.concatenate([vb[r][s][i] for r in range(world)]) for i in range(8))
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            assert abs(z[r]["vlosses"][s] - lref) < 3e-5 * max(1.0, abs(lref)), (s, r, z[r]["vlosses"][s], lref)
    assert np.array_equal(z[0]["w"], z[1]["w"])            # replicas: same all-reduced gradient, same Adam
    N, D = cfg.N, cfg.D
    full = np.zeros((N, D), dtype=np.float32)
    for r in range(world):
        n_r = len(range(r, N, world))
        full[r::world] = z[r]["table"][:n_r]
    d = np.abs(full - ref.table.cpu().numpy())
    assert (d <= 3e-6).mean() > 0.995 and d.max() <= 2.2 * (STEPS + 1 + len(VAR_B)) * 1e-3


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: Tmall-scale (N = 1,529,672 rows, T=20, K=10, D=64, H=128), table row-sharded over
# 2 ranks (real processes, gloo exchange, one GPU), global batch 1024 + an UNEVEN second step (512 + 256).
# ---------------------------------------------------------------------------------------------------
TM_B = (512, 512, 256)          # rank-1 batch sizes per step; rank 0 always trains 512


def _tm_batches(rank):
    from score_amd.synth import make_world
    w, kw = make_world("cfg3")
    kw.pop("batch")
    sizes = (512, 512, 512) if rank == 0 else TM_B
    return w, kw, [w.batch(sizes[s], 40 + 10 * rank + s) for s in range(3)]


def _tm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from score_amd.dist import ShardedSCORE, TorchDistComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, kw, bts = _tm_batches(rank)
    model = ShardedSCORE(seed=21, comm=TorchDistComm(), **kw)     # shard-local initialiser: no full table on any rank
    assert model.backend.m.table.shape[0] == (kw["feature_size"] + world - 1) // world
    losses = []
    for i, bt in enumerate(bts):
        losses.append(model.train(None, bt, 1e-3, 1e-4, keep_prob=1.0, next_batch=bts[i + 1] if i + 1 < len(bts) else None))
    pred, _, _ = model.eval(None, bts[0], 1e-4)
    torch.cuda.synchronize()
    used = np.unique(np.concatenate([np.asarray(a).reshape(-1) for bt in bts for a in bt[:6]]))
    extra = np.random.default_rng(7).choice(kw["feature_size"], 100000, replace=False)
    rows = np.unique(np.concatenate([used, extra]))
    mine = rows[rows % world == rank]
    sel = torch.from_numpy(mine // world).to("cuda")
    np.savez(os.path.join(out_dir, "tm_rank%d.npz" % rank), losses=np.asarray(losses), pred=np.asarray(pred),
             rows=mine, vals=model.backend.m.table[sel].cpu().numpy(), w=model.backend.m.w.cpu().numpy())
    dist.destroy_process_group()


def test_tmall_scale_two_processes_uneven_batches(tmp_path):
    from score_amd.model import SCORE
    world = 2
    mp.spawn(_tm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("tm_rank%d.npz" % r))) for r in range(world)]
    per_rank = [_tm_batches(r) for r in range(world)]
    kw = per_rank[0][1]
    ref = SCORE(seed=21, **kw)                   # same seed: the shards of the workers ARE this table's rows
    for s in range(3):
        cat = tuple(np.concatenate([per_rank[r][2][s][i] for r in range(world)]) for i in range(8))
        assert cat[6].shape[0] == 512 + TM_B[s]
        lref = ref.train(None, cat, 1e-3, 1e-4, keep_prob=1.0)
        for r in range(world):
            assert abs(z[r]["losses"][s] - lref) < 2e-5 * max(1.0, abs(lref)), (s, r, z[r]["losses"][s], lref)
    assert np.array_equal(z[0]["w"], z[1]["w"])
    dw = np.abs(z[0]["w"] - ref.w.cpu().numpy())
    assert np.median(dw) < 1e-6 and dw.max() <= 2.2 * 3 * 1e-3
    tab = ref.table.cpu().numpy()
    for r in range(world):
        d = np.abs(z[r]["vals"] - tab[z[r]["rows"]])
        assert (d <= 3e-6).mean() > 0.999 and d.max() <= 2.2 * 3 * 1e-3, (r, float((d <= 3e-6).mean()), float(d.max()))
        pref, _, _ = ref.eval(None, per_rank[r][2][0], 1e-4)
        dp_ = np.abs(z[r]["pred"] - np.asarray(pref))
        assert np.median(dp_) < 1e-4 and dp_.max() < 2e-3


# ---------------------------------------------------------------------------------------------------
# Round 3: an out-of-range feature id on ONE rank (tf.nn.embedding_lookup would raise, score.py:51-66).  The index plan
# of that rank reports it and routes the id to the dummy row; the global loss is NaN on every rank (it is all-reduced),
# so every rank takes the check together, all-reduces the status words and raises -- nobody is left in a collective.
# ---------------------------------------------------------------------------------------------------
def _badid_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from helpers import batch_tuple, random_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = so.Cfg(*CFG_ARGS, model_type="SCORE")
    model = ShardedSCORE(*CFG_ARGS, comm=TorchDistComm())
    model.backend.m.set_params(so.init_params(cfg, 5))
    good = [random_batch(np.random.default_rng(70 * rank + s), cfg, B) for s in range(3)]
    for b in good:
        b["length"][:] = cfg.T
    msgs = []
    l0 = model.train(None, batch_tuple(good[0]), 1e-3, 1e-3, keep_prob=1.0)
    bad = {k: v.copy() for k, v in good[1].items()}
    if rank == 1:
        bad["item_2hop"][2, 1, 0, 3] = CFG_ARGS[0] + 5          # only rank 1 feeds it
        bad["target_user"][0, 0] = -3
    try:
        model.train(None, batch_tuple(bad), 1e-3, 1e-3, keep_prob=1.0)
        msgs.append("no error")
    except ValueError as e:
        msgs.append(str(e))
    l2 = model.train(None, batch_tuple(good[2]), 1e-3, 1e-3, keep_prob=1.0)       # both ranks go on, in step
    np.savez(os.path.join(out_dir, "bad%d.npz" % rank), l0=l0, l2=l2, msg=np.asarray(msgs[0]),
             finite=bool(torch.isfinite(model.backend.m.table).all() and torch.isfinite(model.backend.m.w).all()))
    dist.destroy_process_group()


def test_bad_id_on_one_rank_raises_on_every_rank(tmp_path):
    world = 2
    mp.spawn(_badid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("bad%d.npz" % r))) for r in range(world)]
    for r in range(world):
        msg = str(z[r]["msg"])
        assert "rank 1" in msg and "(item_2hop)" in msg and "(target_user)" in msg and "rank 0" not in msg, (r, msg)
        assert np.isfinite(float(z[r]["l0"])) and np.isfinite(float(z[r]["l2"])) and bool(z[r]["finite"])
    assert float(z[0]["l2"]) == float(z[1]["l2"])          # the same global loss on both ranks after the error
