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


# ---------------------------------------------------------------------------------------------------
# Pinned by the reference: tests/golden/g5_graph_store.npz holds the documents GraphStore itself wrote
# (tests/golden/make_graph_golden.py imports code/graph_storage.py with in-memory collections).
# ---------------------------------------------------------------------------------------------------
import os

import pytest

G5 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g5_graph_store.npz"))


def _docs(tag, side, field):
    a = G5["%s/%s_%s" % (tag, side, field)]
    n = G5["%s/%s_%s_len" % (tag, side, "2hop" if field == "degrees" else field)]
    return [[a[e, t, :n[e, t]].tolist() for t in range(a.shape[1])] for e in range(a.shape[0])]


def _build(tag, seed=3):
    U, I, S, m1, m2 = [int(x) for x in G5[tag + "/dims"]]
    log = G5[tag + "/log"]
    g = TemporalGraph.from_log(log[:, 0], log[:, 1], log[:, 2], U, I, S, np.arange(1, U + 1)[:, None],
                               np.arange(U + 1, U + I + 1)[:, None], max_1hop=m1, max_2hop=m2, seed=seed)
    return g, (U, I, S, m1, m2)


def test_sparse_log_equals_reference_documents():
    # no cap is hit: 1-hop, 2-hop and degree documents are a pure function of the log -> exact equality
    g, (U, I, S, m1, m2) = _build("sparse")
    for side, n in (("user", U), ("item", I)):
        one, two, deg = _docs("sparse", side, "1hop"), _docs("sparse", side, "2hop"), _docs("sparse", side, "degrees")
        assert max(len(l) for d in one for l in d) <= m1 and max(len(l) for d in two for l in d) <= m2
        gd = g.user_degrees if side == "user" else g.item_degrees
        off2 = (g.user_csr if side == "user" else g.item_csr)["off2"]
        for e in range(n):
            for t in range(S):
                assert g.cell(side, 1, e, t).tolist() == one[e][t], (side, e, t)
                assert g.cell(side, 2, e, t).tolist() == two[e][t], (side, e, t)
                assert gd[off2[e * S + t]:off2[e * S + t + 1]].tolist() == deg[e][t]


@pytest.mark.parametrize("seed", [3, 4])
def test_dense_log_vs_reference_documents(seed):
    # every cap is hit (max_1hop 3, max_2hop 8): cells the reference fills without a random choice must be equal;
    # the others must have the reference's size and draw from the same support
    from collections import Counter
    g, (U, I, S, m1, m2) = _build("dense", seed)
    ref1 = {"user": _docs("dense", "user", "1hop"), "item": _docs("dense", "item", "1hop")}
    ref2 = {"user": _docs("dense", "user", "2hop"), "item": _docs("dense", "item", "2hop")}
    refd = {"user": _docs("dense", "user", "degrees"), "item": _docs("dense", "item", "degrees")}
    log = G5["dense/log"]
    raw = {"user": [[[] for _ in range(S)] for _ in range(U)], "item": [[[] for _ in range(S)] for _ in range(I)]}
    for u, i, t in log.tolist():
        raw["user"][u - 1][t].append(i)
        raw["item"][i - U - 1][t].append(u)
    exact = sampled = 0
    for side, other, n, obase in (("item", "user", I, 1), ("user", "item", U, U + 1)):
        for e in range(n):
            for t in range(S):
                mine1, r1 = g.cell(side, 1, e, t).tolist(), ref1[side][e][t]
                # stored 1-hop list: log order, or a permutation of it when longer than max_1hop (random.shuffle in place)
                assert sorted(mine1) == sorted(r1) == sorted(raw[side][e][t])
                if len(r1) <= m1:
                    assert mine1 == r1 == raw[side][e][t]
                mine2, r2 = g.cell(side, 2, e, t).tolist(), ref2[side][e][t]
                nbs = raw[side][e][t]
                lists = [raw[other][x - obase][t] for x in nbs]
                # the item pass cuts user lists in log order; the user pass cuts item lists the item pass may have permuted
                cut_random = side == "user" and any(len(l) > m1 for l in lists)
                det = len(nbs) <= m1 and not cut_random
                if det:
                    full = [y for l in lists if len(l) > 1 for y in l[:m1]]
                    fdeg = [len(l) for l in lists if len(l) > 1 for _ in l[:m1]]
                    if len(full) <= m2:
                        assert mine2 == r2 == full, (side, e, t)
                        gd = g.user_degrees if side == "user" else g.item_degrees
                        off2 = (g.user_csr if side == "user" else g.item_csr)["off2"]
                        assert gd[off2[e * S + t]:off2[e * S + t + 1]].tolist() == refd[side][e][t] == fdeg
                        exact += 1
                        continue
                    # only the final down-sampling is random: same size, sub-multiset of the same union
                    assert len(mine2) == len(r2) == m2
                    assert not (Counter(mine2) - Counter(full)) and not (Counter(r2) - Counter(full))
                    sampled += 1
                    continue
                support = Counter(y for l in lists if len(l) > 1 for y in l)
                assert len(mine2) <= m2 and len(r2) <= m2
                assert not (Counter(mine2) - support) and not (Counter(r2) - support)
                sampled += 1
    assert exact >= 10 and sampled > 20         # the fixture exercises both kinds of cell


def test_tmall_sample_remap_and_targets():
    # cfg-1 host plumbing: bundled Tmall sample -> one id space (feateng_tmall.py:72-133) -> graph -> target lines
    from score_amd import dataprep as dp
    raw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tmall_sample_log.npz"))["log"]
    r = dp.remap_tmall_log(raw)
    U, I = r["n_users"], r["n_items"]
    assert (U, I) == (62, 5915) and r["vocab_sizes"][2:5] == (450, 1445, 1389)
    assert r["feature_size"] == 1 + sum(r["vocab_sizes"])
    assert r["uid"].min() == 1 and r["uid"].max() == U and r["iid"].min() == U + 1 and r["iid"].max() == U + I
    assert r["t_idx"].min() == 0 and r["t_idx"].max() == 12
    # a raw id always maps to the same row, and rows of different vocabularies never collide
    for col, ids in ((0, r["uid"]), (1, r["iid"])):
        m = {}
        for a, b in zip(raw[:, col].tolist(), ids.tolist()):
            assert m.setdefault(a, b) == b
    ur, ir = r["user_rows"], r["item_rows"]
    assert np.array_equal(ur[:, 0], np.arange(1, U + 1)) and np.array_equal(ir[:, 0], np.arange(U + 1, U + I + 1))
    base = 1 + U + I
    assert ir[:, 1].min() >= base and ir[:, 3].max() < ur[:, 1].min() <= ur[:, 1].max() < ur[:, 2].min()
    assert ur[:, 2].max() == r["feature_size"] - 1
    g = TemporalGraph.from_log(r["uid"], r["iid"], r["t_idx"], U, I, 14, ur, ir)     # TIME_SLICE_NUM_Tmall, graph_storage.py:47
    assert int(g.user_csr["off1"][-1]) == len(raw) == int(g.item_csr["off1"][-1])
    lines = dp.gen_target_lines(r["uid"], r["iid"], r["t_idx"], U, I, 9, 1)
    assert 10 < len(lines) <= U
    for u, its in lines:
        assert len(its) == 2 and U < its[1] <= U + I
        # first item of the prediction slice in LOG order (gen_target.py:112-113 reads the 1-hop database; the 2-hop
        # database's copy of a long list is permuted, see from_log)
        first = r["iid"][(r["uid"] == u) & (r["t_idx"] == 9)][0]
        assert its[0] == first and first in g.cell("user", 1, u - 1, 9)
        assert sum(len(g.cell("user", 1, u - 1, t)) for t in range(9)) > 0


def test_sort_based_build_equals_the_list_build():
    """TemporalGraph.from_log (round 3: stable sort + vectorised 2-hop expansion, random draws in the list build's order)
    against the per-cell Python-list form it replaces (`_from_log_lists`, what rounds 1 - 2 pinned to the reference's
    documents above): every offset, neighbour and degree equal -- also with cells above max_1hop (shuffled in place,
    items before users) and 2-hop lists above max_2hop (down-sampled)"""
    import numpy as np
    from score_amd.graph import TemporalGraph
    for U, I, S, n, m1, m2, seed in ((30, 40, 4, 3000, 10, 100, 1), (200, 50, 6, 20000, 10, 100, 2), (50, 8, 3, 6000, 5, 20, 3),
                                    (600, 900, 11, 9000, 10, 100, 4)):
        rng = np.random.default_rng(seed)
        uid = rng.integers(1, U + 1, n)
        iid = U + 1 + np.minimum((I * rng.random(n) ** 3).astype(np.int64), I - 1)
        t = rng.integers(0, S, n)
        ur, ir = np.zeros((U, 3), np.int32), np.zeros((I, 4), np.int32)
        a = TemporalGraph.from_log(uid, iid, t, U, I, S, ur, ir, m1, m2, seed=5)
        b = TemporalGraph._from_log_lists(uid, iid, t, U, I, S, ur, ir, m1, m2, seed=5)
        for side in ("user_csr", "item_csr"):
            for k in ("off1", "nbr1", "off2", "nbr2"):
                assert np.array_equal(getattr(a, side)[k], getattr(b, side)[k]), (U, side, k)
        assert np.array_equal(a.user_degrees, b.user_degrees) and np.array_equal(a.item_degrees, b.item_degrees)
        over = (np.diff(a.user_csr["off1"]) > m1).sum() + (np.diff(a.item_csr["off1"]) > m1).sum()
        capped = (np.diff(a.user_csr["off2"]) == m2).sum() + (np.diff(a.item_csr["off2"]) == m2).sum()
        assert seed == 4 or (over > 0 and capped > 0)            # the random paths were really taken


def test_graph_file_round_trip(tmp_path):
    """TemporalGraph.save / load: what a db_name of the reference's graph_handler_params resolves to (graph.register_graph)"""
    from score_amd.graph import TemporalGraph, register_graph, resolve_graph
    rng = np.random.default_rng(4)
    n, U, I, S = 400, 12, 20, 5
    uid, iid, t = rng.integers(1, U + 1, n), rng.integers(U + 1, U + I + 1, n), rng.integers(0, S, n)
    urows = np.concatenate([np.arange(1, U + 1)[:, None], rng.integers(40, 50, (U, 1))], 1).astype(np.int32)
    irows = np.concatenate([np.arange(U + 1, U + I + 1)[:, None], rng.integers(50, 60, (I, 2))], 1).astype(np.int32)
    g = TemporalGraph.from_log(uid, iid, t, U, I, S, urows, irows, max_1hop=4, max_2hop=6, seed=2)
    h = TemporalGraph.load(g.save(str(tmp_path / "g.npz")))
    assert (h.U, h.I, h.S, h.Fu, h.Fi) == (g.U, g.I, g.S, g.Fu, g.Fi)
    for a, b in ((g.user_csr, h.user_csr), (g.item_csr, h.item_csr)):
        assert all(np.array_equal(a[k], b[k]) for k in ("off1", "nbr1", "off2", "nbr2"))
    assert np.array_equal(g.user_rows, h.user_rows) and np.array_equal(g.item_rows, h.item_rows)
    assert (g.user_degrees is None) == (h.user_degrees is None)
    if g.user_degrees is not None:
        assert np.array_equal(g.user_degrees, h.user_degrees) and np.array_equal(g.item_degrees, h.item_degrees)
    register_graph("t_2hop", str(tmp_path / "g.npz"))
    assert resolve_graph("t_2hop").U == U and resolve_graph("t_2hop") is resolve_graph("t_2hop")
    with pytest.raises(KeyError):
        resolve_graph("missing_2hop")
