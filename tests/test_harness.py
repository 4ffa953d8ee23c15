"""Ranking metrics vs golden outputs of the reference's own functions (train_score.py:94-142,
extracted and executed by tests/golden/make_metrics_golden.py) and hand-computed cases."""
import math
import os

import numpy as np
import pytest

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


# ---- the training loop's rules (train_score.py:217-258), with a scripted model ------------------------
class _Scripted(object):
    """train() returns 1/step; the validation MRR follows a script, one value per evaluation."""

    def __init__(self, mrrs):
        self.mrrs, self.n_eval, self.n_train, self.saved = list(mrrs), 0, 0, []

    def train(self, sess, batch_data, lr, reg_lambda):
        self.n_train += 1
        return 1.0 / self.n_train

    def save(self, sess, path):
        self.saved.append((self.n_train, path))


def _run(mrrs, n_batches=6, dataset_size=18, bs=4, epochs=6):
    m = _Scripted(mrrs)

    def ev(model, batches, reg_lambda):
        list(batches)
        v = model.mrrs[min(model.n_eval, len(model.mrrs) - 1)]
        model.n_eval += 1
        return 0.0, 0.5, v, v, v, v, v, v, 0.25
    out = h.train_loop(m, lambda: range(n_batches), lambda: range(2), 1e-3, 1e-4, bs, dataset_size, epochs=epochs,
                       save_path="ckpt", evaluate_fn=ev, log=lambda s: None)
    return m, out


def test_train_loop_eval_cadence_and_save_on_best():
    # eval_iter_num = (18 // 3) // (4 / 2) = 3.0: evaluations before step 1 and after steps 3, 6, 9, ...
    m, out = _run([0.1, 0.2, 0.15, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2], n_batches=6)
    assert out["eval_iter_num"] == 3.0 and out["steps"] == 36 and m.n_eval == 13 and not out["early_stopped"]
    assert out["saved_at_steps"] == [3, 9, 12, 15, 18, 21, 24, 27, 30, 33, 36]       # not at step 6 (0.15 < 0.2)
    assert [s for s, _ in m.saved] == out["saved_at_steps"]
    assert out["train_losses"][0] == pytest.approx((1 + 1 / 2 + 1 / 3) / 3)
    assert out["train_losses"][1] == pytest.approx((1 / 4 + 1 / 5 + 1 / 6) / 3)       # reset after every evaluation
    assert out["best_index"] == 12 and out["best_mrr"] == 1.2 and len(out["vali_losses"]) == 13


def test_train_loop_early_stop_rules_only_after_first_epoch():
    # two falls in a row inside epoch 0 do not stop; the first evaluation of epoch 1 (step 9) sees them and stops
    m, out = _run([0.5, 0.4, 0.3, 0.2, 0.1], n_batches=6)
    assert out["early_stopped"] and out["steps"] == 9 and m.n_eval == 4 and out["saved_at_steps"] == []
    assert out["best_index"] == 0
    # plateau rule: two successive gains <= 0.001
    m, out = _run([0.1, 0.3, 0.5, 0.5005, 0.501, 0.9], n_batches=6)
    assert out["early_stopped"] and out["steps"] == 12
    # a gain of more than 0.001 in either of the last two evaluations keeps it going
    m, out = _run([0.1, 0.3, 0.5, 0.5005, 0.6, 0.6005, 0.7, 0.7005, 0.8, 0.9, 1.0, 1.1, 1.2], n_batches=6)
    assert not out["early_stopped"] and out["steps"] == 36


def test_train_loop_rejects_degenerate_cadence():
    with pytest.raises(ValueError):
        _run([0.1], dataset_size=2, bs=200)


def test_train_loop_feeds_lookahead_models_one_batch_ahead():
    class Ahead(_Scripted):
        def __init__(self, mrrs):
            _Scripted.__init__(self, mrrs)
            self.pairs = []

        def train(self, sess, batch_data, lr, reg_lambda, next_batch=None):
            self.pairs.append((batch_data, next_batch))
            return _Scripted.train(self, sess, batch_data, lr, reg_lambda)
    m = Ahead([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7])
    ev = lambda model, batches, reg: (0.0, 0.5, 0, 0, 0, 0, 0, model.mrrs[min(len(model.pairs) // 3, 6)], 0.1)
    h.train_loop(m, lambda: iter([10, 11, 12]), lambda: [], 1e-3, 1e-4, 4, 18, epochs=2, evaluate_fn=ev, log=lambda s: None)
    assert m.pairs == [(10, 11), (11, 12), (12, None)] * 2        # the last batch of an epoch has no successor


def test_step_rollback_restores_the_beta_powers_bit_for_bit():
    """SCOREBASE._rollback_steps (check_ids after suppressed optimizer steps, score.py:51-66 / 101-116): beta1_power and
    beta2_power are chains of fp32 products (adam_advance); k steps back they must be the very bits they were then"""
    import types
    import numpy as np
    from score_amd.model import SCOREBASE, ADAM_B1, ADAM_B2
    m = types.SimpleNamespace(step=0, beta1_power=np.float32(ADAM_B1), beta2_power=np.float32(ADAM_B2))
    hist = {}
    for _ in range(700):
        hist[m.step] = (m.beta1_power, m.beta2_power)
        SCOREBASE.adam_advance(m)
    for k in (1, 3, 64, 699, 700):
        c = types.SimpleNamespace(step=m.step, beta1_power=m.beta1_power, beta2_power=m.beta2_power)
        SCOREBASE._rollback_steps(c, k)
        assert c.step == 700 - k
        assert c.beta1_power == hist[c.step][0] and c.beta2_power == hist[c.step][1]
        assert c.beta1_power.dtype == np.float32 and c.beta2_power.dtype == np.float32
