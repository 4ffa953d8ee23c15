"""The captured training step (SCOREBASE.enable_graph: one hipGraph per batch shape, alpha and the dropout seed read
from device memory -- score_step_scalars_t) against the eager step: same kernels, same arguments, same order, so
losses and every parameter must be BIT-identical, dropout (keep_prob 0.8) included."""
import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from helpers import random_batch, batch_tuple

pytestmark = pytest.mark.gpu


def _pair(cfg, seed=4):
    from score_amd.model import MODELS
    P = so.init_params(cfg, seed)
    out = []
    for _ in range(2):
        m = MODELS[cfg.model_type](cfg.N, cfg.D, cfg.H, cfg.T, cfg.K, cfg.Fu, cfg.Fi, seed=77)
        m.set_params(P)
        out.append(m)
    return out


@pytest.mark.parametrize("model_type", ["SCORE", "RCA", "RIA"])
def test_captured_step_is_bit_identical_to_eager(model_type):
    cfg = so.Cfg(4001, 16, 32, 11, 10, 3, 4, model_type)          # the reference's own shape (train_score.py:15-16, 339-364)
    rng = np.random.default_rng(1)
    eager, graphed = _pair(cfg)
    graphed.enable_graph(True)
    bs = [random_batch(rng, cfg, 200) for _ in range(7)]
    for b in bs:
        b["length"] = np.full(200, 9, dtype=np.int32)             # train split: 9 of 11 slices (graph_loader.py:382)
    other = random_batch(rng, cfg, 100)                           # a second shape: its own capture
    seq = [bs[0], bs[1], bs[2], other, bs[3], other, bs[4], other, bs[5], other, bs[6], bs[0]]
    for i, b in enumerate(seq):
        le = eager.train(None, batch_tuple(b), 1e-3, 1e-4)        # keep_prob 0.8: dropout masks from the step's seed
        lg = graphed.train(None, batch_tuple(b), 1e-3, 1e-4)
        assert le == lg, (i, le, lg)
        if i == 6:                                                # evaluation between captured steps
            pe, _, _ = eager.eval(None, batch_tuple(bs[1]), 1e-4)
            pg, _, _ = graphed.eval(None, batch_tuple(bs[1]), 1e-4)
            assert pe == pg
    assert len([v for v in graphed._graphs.values() if isinstance(v, tuple)]) == 2       # both shapes were captured
    assert torch.equal(eager.w, graphed.w) and torch.equal(eager.table, graphed.table)
    assert torch.equal(eager.table_m, graphed.table_m) and torch.equal(eager.w_v, graphed.w_v)
    assert eager.step == graphed.step == len(seq) and eager.beta1_power == graphed.beta1_power
    # a different lr takes effect without a new capture (alpha lives in device memory)
    n_graphs = len(graphed._graphs)
    assert eager.train(None, batch_tuple(bs[2]), 5e-4, 1e-4) == graphed.train(None, batch_tuple(bs[2]), 5e-4, 1e-4)
    assert len(graphed._graphs) == n_graphs and torch.equal(eager.w, graphed.w)
    # switching it off returns to the eager path
    graphed.enable_graph(False)
    assert eager.train(None, batch_tuple(bs[3]), 1e-3, 1e-4) == graphed.train(None, batch_tuple(bs[3]), 1e-3, 1e-4)


def test_captured_step_with_device_loader_batches():
    import os
    from score_amd import dataprep as dp
    from score_amd.graph import DeviceGraphLoader
    from score_amd.model import SCORE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    raw = np.load(os.path.join(root, "tests", "golden", "tmall_sample_log.npz"))["log"]
    g, r, targets = dp.tmall_pipeline(raw)
    c = dp.TMALL
    T, K = c["time_slice_num"] - 1, c["obj_per_time_slice"]
    args = (r["feature_size"], c["eb_dim"], c["hidden_size"], T, K, c["user_fnum"], c["item_fnum"])
    cfg = so.Cfg(*args, model_type="SCORE")
    P = so.init_params(cfg, 2)
    a, b = SCORE(*args, seed=5), SCORE(*args, seed=5)
    a.set_params(P); b.set_params(P)
    b.enable_graph(True)
    g.to_device()
    for epoch in range(4):
        la = [a.train(None, bt, 1e-3, 1e-4) for bt in DeviceGraphLoader(g, 16, targets["train"], 0, 9, 1, T, K)]
        lb = [b.train(None, bt, 1e-3, 1e-4) for bt in DeviceGraphLoader(g, 16, targets["train"], 0, 9, 1, T, K)]
        assert la == lb
    assert torch.equal(a.w, b.w) and torch.equal(a.table, b.table)


def test_async_captured_steps_keep_their_own_scalars():
    """with a captured step the host runs many steps ahead of the GPU (train_async, no sync), and every
    step's alpha / dropout seed travels through a pinned staging slot.  One shared slot was rewritten for step t+1
    before step t's queued copy had read it; now a ring of slots, each guarded by the event of its copy.  Different
    learning rates per step (alpha differs visibly) and dropout on: 40 unsynchronised replays must equal the eager run."""
    cfg = so.Cfg(3001, 16, 32, 6, 5, 3, 4, "SCORE")
    rng = np.random.default_rng(8)
    eager, graphed = _pair(cfg)
    graphed.enable_graph(True)
    bs = [batch_tuple(random_batch(rng, cfg, 64)) for _ in range(4)]
    dbs_e = [eager.device_batch(b) for b in bs]
    dbs_g = [graphed.device_batch(b) for b in bs]
    lrs = [1e-3 * (1 + (i % 7)) for i in range(44)]
    le = [eager.train_async(dbs_e[i % 4], lr, 1e-4).clone() for i, lr in enumerate(lrs)]
    lg = []
    torch.cuda.synchronize()
    pad = torch.zeros((1 << 26,), device="cuda")
    for i, lr in enumerate(lrs):
        if i == 4:
            for _ in range(30):          # a backlog on the stream: the replays below are queued far ahead of execution
                pad.add_(1.0)
        lg.append(graphed.train_async(dbs_g[i % 4], lr, 1e-4).clone())
    torch.cuda.synchronize()
    assert len([v for v in graphed._graphs.values() if isinstance(v, tuple)]) == 1
    assert torch.equal(torch.stack(le), torch.stack(lg))
    assert torch.equal(eager.w, graphed.w) and torch.equal(eager.table, graphed.table) and torch.equal(eager.w_m, graphed.w_m)
