"""Host-side temporal graph construction (row f2) -- semantics of graph_storage.py:93-246 restated:
1-hop lists in log order; 2-hop = slice-local 1-hop lists of the node's neighbours (degree > 1 only,
each cut to max_1hop), capped at max_2hop.  CPU only."""
import numpy as np

from score_amd.graph import TemporalGraph


def cell(csr, key_off, key_nbr, e, t, S):
    o = csr[key_off]
    return csr[key_nbr][o[e * S + t]:o[e * S + t + 1]].tolist()


def test_from_log_semantics():
    U, I, S = 4, 5, 3
    # (uid, iid, t)
    log = [(1, 5, 0), (1, 6, 0), (2, 5, 0), (3, 5, 0), (2, 7, 1), (2, 7, 1), (4, 9, 2), (1, 9, 2), (1, 8, 2)]
    uid, iid, t = zip(*log)
    g = TemporalGraph.from_log(uid, iid, t, U, I, S, np.arange(1, U + 1)[:, None], np.arange(U + 1, U + I + 1)[:, None])
    assert cell(g.user_csr, "off1", "nbr1", 0, 0, S) == [5, 6]            # log order
    assert cell(g.item_csr, "off1", "nbr1", 0, 0, S) == [1, 2, 3]
    assert cell(g.user_csr, "off1", "nbr1", 1, 1, S) == [7, 7]            # repeats are kept
    # user 1, slice 0: neighbours 5 (degree 3 > 1 -> its users [1,2,3]) and 6 (degree 1 -> skipped)
    assert cell(g.user_csr, "off2", "nbr2", 0, 0, S) == [1, 2, 3]
    # item 5, slice 0: users 1 (degree 2 -> [5,6]), 2 (degree 1, skipped), 3 (degree 1, skipped)
    assert cell(g.item_csr, "off2", "nbr2", 0, 0, S) == [5, 6]
    # slice-local: user 1's slice-2 items are 9 (users [4,1]) and 8 (degree 1)
    assert cell(g.user_csr, "off2", "nbr2", 0, 2, S) == [4, 1]
    assert cell(g.user_csr, "off2", "nbr2", 3, 1, S) == []


def test_from_log_caps():
    U, I, S = 30, 3, 1
    uid = list(range(1, 31)) * 2
    iid = [31] * 30 + [32] * 30
    g = TemporalGraph.from_log(uid, iid, [0] * 60, U, I, S, np.arange(1, U + 1)[:, None],
                               np.arange(U + 1, U + I + 1)[:, None], max_1hop=10, max_2hop=15)
    two = cell(g.item_csr, "off2", "nbr2", 0, 0, S)      # item 31: 10 sampled users x their 2 items
    assert len(two) == 15 and set(two) <= {31, 32}
    assert len(cell(g.user_csr, "off2", "nbr2", 0, 0, S)) == 15       # 2 items x first 10 users, capped at 15
    assert len(cell(g.item_csr, "off1", "nbr1", 0, 0, S)) == 30       # 1-hop lists are not capped in the store


def test_to_device_requires_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = TemporalGraph.from_log([1], [2], [0], 1, 1, 1, np.array([[1]]), np.array([[2]]))
    with pytest.raises(RuntimeError):
        g.to_device()
