"""BASELINE.json configs[0]: the reference's bundled Tmall sample through the whole pipeline on the MI355X path --
raw log (tests/golden/tmall_sample_log.npz: the CSV's columns) -> id remap (feateng_tmall.py:72-133) -> temporal
graph (graph_storage.py:93-246) -> target lines (gen_target.py:99-121) -> DeviceGraphLoader (graph_loader.py) ->
SCORE.train / eval under the training loop's rules (train_score.py:165-275), with the Tmall constants of
train_score.py:339-364 (T = 11, K = 10, D = 16, H = 32, Fu = 3, Fi = 4, pred slices 9 / 10 / 11, eval at 1 + 99).
In the reference this configuration is "TF1.x CPU (plumbing, no GPU)": here the same batches also go through the
CPU oracle, which must agree with the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import score_oracle as so

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pipeline():
    from score_amd import dataprep as dp
    raw = np.load(os.path.join(ROOT, "tests", "golden", "tmall_sample_log.npz"))["log"]
    g, r, targets = dp.tmall_pipeline(raw)
    g.to_device()
    return dp.TMALL, g, r, targets


def _loader(c, g, targets, split, batch):
    from score_amd.graph import DeviceGraphLoader
    T = c["time_slice_num"] - c["start_time"] - 1
    pt = c["pred_time_" + split]
    neg = c["train_neg"] if split == "train" else c["test_neg"]
    return DeviceGraphLoader(g, batch, targets[split], c["start_time"], pt, neg, T, c["obj_per_time_slice"])


def test_train_loop_on_the_sample():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cfg1_tmall_sample as cfg1
    logs = []
    model, out = cfg1.run(batch=8, lr=1e-3, reg_lambda=1e-4, log=logs.append)
    assert out["lines"] == {"train": 41, "validation": 43, "test": 44} and out["feature_size"] == 9274
    assert out["eval_iter_num"] == 3.0                         # (41 // 3) // (8 / 2), train_score.py:217
    assert out["steps"] >= 3 and len(out["vali_mrrs"]) == 1 + out["steps"] // 3
    assert len(logs) == len(out["vali_mrrs"]) and logs[0].startswith("STEP 0  LOSS TRAIN: NULL")
    for k in ("vali_ndcgs_5", "vali_ndcgs_10", "vali_hrs_1", "vali_hrs_5", "vali_hrs_10", "vali_mrrs"):
        assert all(0.0 <= v <= 1.0 for v in out[k]), k
    assert all(np.isfinite(v) for v in out["train_losses"] + out["vali_losses"])
    t = out["test"]
    assert 0.0 < t["logloss"] < 5.0 and 0.0 <= t["auc"] <= 1.0 and 0.0 < t["mrr"] <= 1.0
    assert t["hr_1"] <= t["hr_5"] <= t["hr_10"] and t["ndcg_5"] <= t["ndcg_10"]


def test_fifty_steps_at_the_reference_batch_size_fit_the_sample(pipeline):
    # train_score.py:372 batch 200 (the 41 target lines make one short batch of 82 per epoch): >= 50 steps drive
    # the training loss down and the ranking of the training positives up -- the path learns on real data
    from score_amd import harness
    from score_amd.model import SCORE
    c, g, r, targets = pipeline
    T = c["time_slice_num"] - c["start_time"] - 1
    m = SCORE(r["feature_size"], c["eb_dim"], c["hidden_size"], T, c["obj_per_time_slice"], c["user_fnum"], c["item_fnum"])
    losses = []
    for epoch in range(60):
        for b in _loader(c, g, targets, "train", 200):
            assert b.B == 82 and b.active_slices == 9
            losses.append(m.train(None, b, 1e-3, 1e-4))
    assert len(losses) == 60 and np.mean(losses[-5:]) < 0.6 * np.mean(losses[:5])
    res = harness.evaluate_device(m, _loader(c, g, targets, "train", 200), 1e-4, neg_sample_num=1)
    assert res[1] > 0.95                                       # AUC on the fitted training pairs
    # the reference-shaped evaluate() accepts the device loader's batches too and agrees with the device metrics
    dev = harness.evaluate_device(m, _loader(c, g, targets, "validation", 100), 1e-4)
    host = harness.evaluate(m, _loader(c, g, targets, "validation", 100), 1e-4)
    assert np.allclose(dev, host, rtol=0, atol=2e-6), (dev, host)


def test_hip_path_equals_oracle_on_real_batches(pipeline):
    # the same assembled batches (real id distribution: 62 users, hot sellers / brands, many dummy slices) through
    # the CPU restatement: predictions within 1e-4, three training steps' losses within 1e-5 relative
    from score_amd.model import SCORE
    c, g, r, targets = pipeline
    T = c["time_slice_num"] - c["start_time"] - 1
    args = (r["feature_size"], c["eb_dim"], c["hidden_size"], T, c["obj_per_time_slice"], c["user_fnum"], c["item_fnum"])
    cfg = so.Cfg(*args, model_type="SCORE")
    P = so.init_params(cfg, 3)
    m = SCORE(*args)
    m.set_params(P)
    om = so.OracleModel(*args, model_type="SCORE", params={k: v.copy() for k, v in P.items()})
    train = [tuple(t.cpu().numpy() for t in b.tensors) for b in _loader(c, g, targets, "train", 32)]
    vali = [tuple(t.cpu().numpy() for t in b.tensors) for b in _loader(c, g, targets, "validation", 100)][:3]
    assert len(train) == 3 and train[-1][6].shape[0] == 18      # 41 lines x 2 in batches of 32: short last batch
    for b in vali:
        pg, lg, _ = m.eval(None, b, 1e-4)
        po, lo, _ = om.eval(None, b, 1e-4)
        assert lg == lo and np.abs(np.asarray(pg) - np.asarray(po)).max() < 1e-4
    for b in train:
        l_g = m.train(None, b, 1e-3, 1e-4, keep_prob=1.0)
        l_o = om.train(None, b, 1e-3, 1e-4, keep_prob=1.0)
        assert abs(l_g - l_o) < 1e-5 * max(1.0, abs(l_o))
    pg, _, _ = m.eval(None, vali[0], 1e-4)
    po, _, _ = om.eval(None, vali[0], 1e-4)
    d = np.abs(np.asarray(pg) - np.asarray(po))
    assert np.median(d) < 1e-4 and d.max() < 2e-3
