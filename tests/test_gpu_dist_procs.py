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
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.asarray(losses), pred=np.asarray(pred),
             table=model.backend.m.table.cpu().numpy(), w=model.backend.m.w.cpu().numpy())
    dist.destroy_process_group()


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
    assert np.array_equal(z[0]["w"], z[1]["w"])            # replicas: same all-reduced gradient, same Adam
    N, D = cfg.N, cfg.D
    full = np.zeros((N, D), dtype=np.float32)
    for r in range(world):
        n_r = len(range(r, N, world))
        full[r::world] = z[r]["table"][:n_r]
    d = np.abs(full - ref.table.cpu().numpy())
    assert (d <= 3e-6).mean() > 0.999 and d.max() <= 2.2 * STEPS * 1e-3
    for r in range(world):
        pref, _, _ = ref.eval(None, batch_tuple(per_rank[r][0]), 1e-3)
        assert np.abs(z[r]["pred"] - np.asarray(pref)).max() < 1e-4
