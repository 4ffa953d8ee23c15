"""G5: golden 1-hop / 2-hop / degree documents from the REFERENCE graph store (code/graph_storage.py
GraphStore.construct_coll_1hop / construct_coll_2hop, imported in the build container).  No MongoDB server:
the two databases are replaced by in-memory fakes with insert_many / find, returning copies the way a
database does.  Commits only data: the behaviour logs fed in and the documents the reference wrote, as
padded arrays.

Run (container only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_graph_golden.py"""
import copy
import os
import random
import sys
import tempfile

import numpy as np

REF = "/root/reference/code"
HERE = os.path.dirname(os.path.abspath(__file__))


class FakeColl(object):
    def __init__(self):
        self.docs = []

    def insert_many(self, docs):
        self.docs += [copy.deepcopy(d) for d in docs]

    def find(self, q):
        assert q == {}
        return [copy.deepcopy(d) for d in self.docs]


class FakeDB(dict):
    def __missing__(self, k):
        self[k] = FakeColl()
        return self[k]


def pad(docs, field, S, width):
    out = np.zeros((len(docs), S, width), dtype=np.int32)
    lens = np.zeros((len(docs), S), dtype=np.int32)
    for e, d in enumerate(docs):
        for t in range(S):
            l = d[field][t]
            assert len(l) <= width, (field, len(l), width)
            out[e, t, :len(l)] = l
            lens[e, t] = len(l)
    return out, lens


def main():
    sys.path.insert(0, REF)
    import graph_storage as gs
    rng = np.random.Generator(np.random.PCG64(5))
    blob = {}
    # (tag, users, items, slices, rows, max_1hop, max_2hop, users/collection, items/collection)
    cases = (("dense", 12, 15, 4, 260, 3, 8, 5, 4),        # every cap is hit: shuffles, per-neighbour cuts, down-sampling
             ("sparse", 30, 40, 5, 200, 10, 100, 200, 500))  # no cap is hit: the documents are a pure function of the log
    for tag, U, I, S, n, m1, m2, upc, ipc in cases:
        uid = rng.integers(1, U + 1, n)
        iid = rng.integers(U + 1, U + I + 1, n)
        t = rng.integers(0, S, n)
        tmp = tempfile.mkdtemp()
        path = os.path.join(tmp, "log.csv")
        with open(path, "w") as f:
            for a, b, c in zip(uid.tolist(), iid.tolist(), t.tolist()):
                f.write("%d,%d,_,%d\n" % (a, b, c))
        store = gs.GraphStore(path, user_per_collection=upc, item_per_collection=ipc, start_time=0, max_1hop=m1,
                              max_2hop=m2, user_num=U, item_num=I, db_1hop="x", db_2hop="y", time_slice_num=S)
        store.db_1hop, store.db_2hop = FakeDB(), FakeDB()
        random.seed(11)                      # graph_storage.py:12
        np.random.seed(11)
        store.construct_coll_1hop()
        store.construct_coll_2hop()
        n_uc = (U + upc - 1) // upc
        n_ic = (I + ipc - 1) // ipc
        udocs = [d for c in range(n_uc) for d in store.db_2hop["user_%d" % c].docs][:U]
        idocs = [d for c in range(n_ic) for d in store.db_2hop["item_%d" % c].docs][:I]
        assert [d["uid"] for d in udocs] == list(range(1, U + 1))
        assert [d["iid"] for d in idocs] == list(range(U + 1, U + I + 1))
        w1 = max(len(l) for d in udocs + idocs for l in d["1hop"])
        blob[tag + "/dims"] = np.asarray([U, I, S, m1, m2])
        blob[tag + "/log"] = np.stack([uid, iid, t], axis=1).astype(np.int32)
        for nm, docs in (("user", udocs), ("item", idocs)):
            blob["%s/%s_1hop" % (tag, nm)], blob["%s/%s_1hop_len" % (tag, nm)] = pad(docs, "1hop", S, w1)
            blob["%s/%s_2hop" % (tag, nm)], blob["%s/%s_2hop_len" % (tag, nm)] = pad(docs, "2hop", S, m2)
            blob["%s/%s_degrees" % (tag, nm)], _ = pad(docs, "degrees", S, m2)
    out = os.path.join(HERE, "g5_graph_store.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
