"""world_size-2 gloo (CPU) test of the sharded data-parallel path: score_amd/dist.py's request /
row / gradient routing, dense all-reduce and global-batch loss scaling, against ONE oracle model
trained on the concatenated batch.  Compute is the oracle-backed CpuBackend (tests/cpu_backend.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_type, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from cpu_backend import CpuBackend
    from helpers import random_batch, batch_tuple
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg_args = (203, 4, 8, 3, 3, 3, 4)            # odd N: the last shard is padded
    cfg = so.Cfg(*cfg_args, model_type=model_type)
    params = so.init_params(cfg, 5)
    rng = np.random.default_rng(100 + rank)
    batches = [random_batch(rng, cfg, 6) for _ in range(2)]
    be = CpuBackend(rank, world, model_type, cfg_args, params)
    model = ShardedSCORE(*cfg_args, comm=TorchDistComm(), backend=be, model_type=model_type)
    bts = [batch_tuple(b) for b in batches]
    # the next batch's index-only phase is run ahead, inside the current step
    losses = [model.train(None, bts[0], 1e-3, 1e-3, keep_prob=1.0, next_batch=bts[1])]
    losses.append(model.train(None, bts[1], 1e-3, 1e-3, keep_prob=1.0))
    pred, label, eloss = model.eval(None, batch_tuple(batches[0]), 1e-3)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.asarray(losses), shard=be.full_table_part(),
             pred=np.asarray(pred), eloss=eloss, **{"dense/" + k: v for k, v in be.dense.items()},
             **{"b%d/%s" % (i, k): v for i, b in enumerate(batches) for k, v in b.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["SCORE", "RCA"])
def test_two_rank_sharded_training_matches_single_model(tmp_path, model_type):
    from oracle import score_oracle as so
    from helpers import NAMES
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, model_type, str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    cfg_args = (203, 4, 8, 3, 3, 3, 4)
    cfg = so.Cfg(*cfg_args, model_type=model_type)
    ref = so.OracleModel(*cfg_args, model_type=model_type, params=so.init_params(cfg, 5))
    for i in range(2):
        cat = tuple(np.concatenate([z[r]["b%d/%s" % (i, n)] for r in range(world)]) for n in NAMES)
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            assert abs(z[r]["losses"][i] - lref) < 1e-5 * max(1.0, abs(lref)), (i, r)
    # replicated dense variables: identical on both ranks and equal to the single model's
    # (the bias of the last attention layer is softmax-shift-invariant: its true gradient is 0, so
    #  Adam amplifies rounding noise to +-lr there; it is only required to agree across ranks)
    shift_invariant = {"SCORE": "dense_5/bias", "RCA": "dense_3/bias"}[model_type]
    for k, v in ref.params.items():
        if k == "emb_mtx":
            continue
        assert np.array_equal(z[0]["dense/" + k], z[1]["dense/" + k]), k
        if k != shift_invariant:
            assert np.allclose(z[0]["dense/" + k], v, rtol=0, atol=2e-6), k
    # table: shard r holds rows r, r+G, ... ; padded tail rows stay zero
    N = cfg.N
    full = np.zeros((N, cfg.D), dtype=np.float32)
    for r in range(world):
        part = z[r]["shard"]
        n_r = len(range(r, N, world))
        full[r::world] = part[:n_r]
        assert not part[n_r:].any()
    want = ref.params["emb_mtx"].copy()
    want[0] = 0
    assert np.allclose(full, want, rtol=0, atol=2e-6)
    # eval on rank r's first batch == the single model evaluated on that batch
    for r in range(world):
        b = tuple(z[r]["b0/%s" % n] for n in NAMES)
        pref, _, lref = ref.eval(None, b, 1e-3)
        assert np.allclose(z[r]["pred"], pref, atol=1e-5)
        assert abs(float(z[r]["eloss"]) - lref) < 1e-5
