"""Ranking metrics vs golden outputs of the reference's own functions (train_score.py:94-142,
extracted and executed by tests/golden/make_metrics_golden.py) and hand-computed cases."""
import math
import os

import numpy as np

from score_amd import harness as h

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_golden.npz"))


def test_ranking_quality_matches_reference_outputs():
    for case in range(3):
        got = h.get_ranking_quality(Z["c%d/preds" % case].tolist(), Z["c%d/iids" % case].tolist())
        assert np.allclose(got, Z["c%d/quality" % case], rtol=0, atol=1e-12), case
        assert abs(h.get_ndcg(Z["c%d/preds" % case], Z["c%d/iids" % case]) - float(Z["c%d/ndcg5" % case])) < 1e-12


def test_scalar_helpers_match_reference_outputs():
    rl = Z["scalar/ranklist"].tolist()
    for t in range(10):
        for j, k in enumerate((1, 5, 10)):
            assert abs(h.getNDCG_at_K(rl, t, k) - Z["scalar/ndcg"][t, j]) < 1e-15
            assert h.getHR_at_K(rl, t, k) == Z["scalar/hr"][t, j]
    for t in range(11):
        assert abs(h.getMRR(rl, t) - Z["scalar/mrr"][t]) < 1e-15


def test_hand_computed():
    # one line of 100: positive scored 3rd best -> rank index 2
    preds = np.linspace(0, 0.5, 100)
    preds[0] = 0.4925
    ids = np.arange(100) + 10
    q = h.get_ranking_quality(preds, ids)
    assert abs(q[0] - math.log(2) / math.log(4)) < 1e-12 and q[2] == 0 and q[3] == 1 and abs(q[5] - 1 / 3) < 1e-12


def test_evaluate_with_stub_model():
    class Stub(object):
        def eval(self, sess, batch_data, reg_lambda):
            n = len(batch_data[6])
            return np.linspace(0.9, 0.1, n).tolist(), list(batch_data[6]), 0.5
    batch = ([0] * 100, 0, 0, 0, 0, [[i + 1, 0] for i in range(100)], [1] + [0] * 99, 0)
    res = h.evaluate(Stub(), [batch, batch], 1e-4)
    assert len(res) == 9 and res[1] == 1.0 and res[4] == 1.0 and res[8] == 0.5


def test_active_slices_rule():
    # score_batch_t.active_slices = longest sample, clipped to [1, T]; 0 (= all T) when nothing is masked or the
    # model has the skip switched off
    from score_amd.model import active_slices

    class Cfg(object):
        max_time_len = 11

    class M(object):
        cfg = Cfg()
        skip_masked_slices = True
    assert active_slices(M(), 9) == 9 and active_slices(M(), 11) == 0 and active_slices(M(), 15) == 0
    assert active_slices(M(), 0) == 1 and active_slices(M(), -3) == 1
    M.skip_masked_slices = False
    assert active_slices(M(), 9) == 0
