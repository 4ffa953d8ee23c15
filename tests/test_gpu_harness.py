"""On-device ranking metrics (score_ranking_quality) against the golden outputs of the reference's own
metric functions (tests/golden/make_metrics_golden.py, train_score.py:104-142) and against the host harness."""
import os

import numpy as np
import pytest
import torch

from score_amd import harness as h

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_golden.npz"))


def test_device_ranking_quality_matches_reference_outputs():
    for case in range(3):
        preds = np.asarray(Z["c%d/preds" % case], dtype=np.float32)
        iids = np.asarray(Z["c%d/iids" % case], dtype=np.int32)
        # the golden values were computed on the float64 predictions; ranks only depend on their order, which the
        # float32 copies keep unless two distinct scores collapse -- compare with the host harness on the same fp32
        want = h.get_ranking_quality(preds.astype(np.float64).tolist(), iids.tolist())
        got = h.ranking_quality_device(torch.from_numpy(preds).cuda(), torch.from_numpy(iids).cuda())
        assert np.allclose(got, want, rtol=0, atol=2e-6), case
        if np.array_equal(np.argsort(preds.reshape(-1, 100), 1), np.argsort(Z["c%d/preds" % case].reshape(-1, 100), 1)):
            assert np.allclose(got, Z["c%d/quality" % case], rtol=0, atol=2e-6), case


def test_device_ranks_ties_and_repeated_positive_id():
    rng = np.random.default_rng(4)
    n, per = 257, 10
    preds = rng.integers(0, 4, (n, per)).astype(np.float32) / 4          # many ties
    iids = rng.integers(1, 6, (n, per)).astype(np.int32)                 # the positive's id recurs among the negatives
    res, ranks = h.ranking_quality_device(torch.from_numpy(preds).cuda(), torch.from_numpy(iids).cuda(),
                                          neg_sample_num=per - 1, return_ranks=True)
    want_r = h._ranked(preds.reshape(-1).tolist(), iids.reshape(-1).tolist(), per)
    assert np.array_equal(ranks.cpu().numpy(), want_r)
    want = h.get_ranking_quality(preds.reshape(-1).tolist(), iids.reshape(-1).tolist(), per - 1)
    assert np.allclose(res, want, rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        h.ranking_quality_device(torch.zeros(7).cuda(), torch.zeros(7, dtype=torch.int32).cuda(), 3)


def test_device_auc_logloss_match_sklearn():
    from sklearn.metrics import log_loss, roc_auc_score
    rng = np.random.default_rng(8)
    for n, ties in ((5000, False), (3001, True), (64, True)):
        p = rng.random(n).astype(np.float32) * 0.98 + 0.01
        if ties:
            p = np.round(p, 1 if n < 100 else 2).astype(np.float32).clip(0.01, 0.99)   # long runs of equal scores
        y = (rng.random(n) < p).astype(np.int32)
        y[:2] = [0, 1]
        auc, ll = h.auc_logloss_device(torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda())
        assert abs(auc - roc_auc_score(y, p.astype(np.float64))) < 1e-12
        assert abs(ll - log_loss(y, p.astype(np.float64))) < 1e-12
    auc, _ = h.auc_logloss_device(torch.full((8,), 0.3).cuda(), torch.ones(8, dtype=torch.int32).cuda())
    assert np.isnan(auc)                                       # one class only


def test_evaluate_device_equals_host_evaluate():
    from score_amd.synth import make_world
    from score_amd.model import SCORE
    w, kw = make_world("cfg2")
    kw.pop("batch")
    m = SCORE(seed=3, **kw)
    neg, lines = 9, 12
    batches = []
    for i in range(3):
        b = list(w.batch(lines * (neg + 1), 40 + i))
        b[6] = (np.arange(lines * (neg + 1)) % (neg + 1) == 0).astype(np.int32)     # one positive per line
        batches.append(tuple(b))
    host = h.evaluate(m, [tuple(a.tolist() for a in b) for b in batches], 1e-4, neg_sample_num=neg)
    dev = h.evaluate_device(m, batches, 1e-4, neg_sample_num=neg)
    assert np.allclose(host, dev, rtol=1e-5, atol=2e-6)
