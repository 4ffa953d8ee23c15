"""The reference's other two data-set shapes at model level:
  CCMR    train_score.py:285-311  T = 40 (train length 38), K = 10, Fu = 1, Fi = 5, N = 5,405,586, D = 16, H = 32, B = 200
  Taobao  train_score.py:311-338  T = 8  (train length 6),  K = 10, Fu = 1, Fi = 2, N = 5,042,754, D = 16, H = 32, B = 200
(EMBEDDING_SIZE / HIDDEN_SIZE :15-16, batch sizes :372).  The HIP path runs on the full table; the oracle on the compacted
id space of the batch (tests/test_gpu_cfg5.py explains why that is the same computation): predictions within 1e-4, loss
1e-5 relative, gradients 2e-4 of the tensor's maximum, one TF-Adam step; plus the size-independent properties of a full
B = 200 batch on the full table."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_cfg5 import _check_vs_oracle, _model, full_batch_properties      # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ccmr():
    out = _model("ccmr_default")
    yield out
    del out
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def taobao_default():
    out = _model("taobao_default")
    yield out
    del out
    torch.cuda.empty_cache()


def test_ccmr_constants():
    from score_amd.synth import make_world, CONFIGS
    U, I, T, K, D, H, B, Fu, Fi, iv, uv = CONFIGS["ccmr_default"]
    # FEAT_SIZE_CCMR, TIME_SLICE_NUM_CCMR - START_TIME_CCMR - 1, OBJ_PER_TIME_SLICE_CCMR (train_score.py:24-32, 288-310)
    assert 1 + U + I + sum(iv) + sum(uv) == 1 + 4920695 + 190129 + (80171 + 1) + (213481 + 1) + (62 + 1) + (1043 + 1) == 5405586
    assert (T, K, D, H, Fu, Fi) == (40, 10, 16, 32, 1, 5)
    w, kw = make_world("ccmr_default")
    assert kw["feature_size"] == 5405586
    b = w.batch(4, 0)
    assert b[0].shape == (4, 40, 10, 5) and b[1].shape == (4, 40, 10, 1) and int(b[7][0]) == 38      # pred_time_train 38
    assert int(max(a.max() for a in b[:6])) < 5405586


def test_ccmr_shape_vs_oracle(ccmr):
    w, kw, B, m = ccmr
    _check_vs_oracle(w, kw, m, 64, ragged=False)


def test_ccmr_shape_vs_oracle_ragged_lengths(ccmr):
    from score_amd.model import SCORE
    w, kw, B, _ = ccmr
    _check_vs_oracle(w, kw, SCORE(seed=8, **kw), 40, ragged=True)


def test_ccmr_full_batch_properties(ccmr):
    full_batch_properties(*ccmr)


@pytest.mark.parametrize("mt", ["RIA", "RCA", "RRN"])
def test_ccmr_shape_other_model_types(ccmr, mt):
    """the ablations at T = 40 / Fi = 5: predictions and loss against the oracle on the compacted ids"""
    from score_amd.model import MODELS
    from test_gpu_cfg5 import compact_oracle
    from oracle import score_oracle as so
    w, kw, B, _ = ccmr
    m = MODELS[mt](seed=4, **kw)
    b = w.batch(32, 5)
    om, rb, uniq = compact_oracle(m, kw, b)
    om = so.OracleModel(len(uniq), kw["eb_dim"], kw["hidden_size"], kw["max_time_len"], kw["obj_per_time_slice"],
                        kw["user_fnum"], kw["item_fnum"], mt, params=om.params)
    pg, _, lg = m.eval(None, b, 1e-4)
    po, _, lo = om.eval(None, rb, 1e-4)
    assert np.abs(np.asarray(pg) - np.asarray(po)).max() < 1e-4 and abs(lg - lo) < 1e-5 * max(1.0, abs(lo))
    l_g = m.train(None, b, 1e-3, 1e-4, keep_prob=1.0)
    l_o = om.train(None, rb, 1e-3, 1e-4, keep_prob=1.0)
    assert abs(l_g - l_o) < 1e-5 * max(1.0, abs(l_o))
    del m
    torch.cuda.empty_cache()


def test_taobao_default_shape_vs_oracle(taobao_default):
    w, kw, B, m = taobao_default
    _check_vs_oracle(w, kw, m, 64, ragged=True)
    full_batch_properties(w, kw, B, m)
