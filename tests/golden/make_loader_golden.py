"""G4: golden batches from the REFERENCE loader (code/score/graph_loader.py GraphHandler, imported in
the build container; needs no MongoDB server: its collections are replaced by in-memory fakes holding
the documents graph_storage.py would have written).  Commits only data: the documents (as padded
arrays), the feature dictionaries and the [T][K][F] lists the reference produced.

Run (container only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_loader_golden.py"""
import copy
import os
import pickle
import sys
import tempfile

import numpy as np

REF = "/root/reference/code/score"
HERE = os.path.dirname(os.path.abspath(__file__))


class FakeColl(object):
    def __init__(self, docs, key):
        self.docs = {d[key]: d for d in docs}
        self.key = key

    def find(self, q):
        return [copy.deepcopy(self.docs[q[self.key]])]     # the reference mutates the lists it gets


def make_docs(rng, n, first_id, nbr_lo, nbr_hi, S, max2, key):
    docs = []
    for e in range(n):
        d = {key: first_id + e, "1hop": [], "2hop": [], "degrees": []}
        for t in range(S):
            l1 = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7, 9]))
            l2 = 0 if l1 == 0 else int(rng.integers(0, max2 + 1))
            d["1hop"].append(rng.integers(nbr_lo, nbr_hi, l1).tolist())
            d["2hop"].append(rng.integers(first_id, first_id + n, l2).tolist())
            d["degrees"].append(rng.integers(2, 9, l2).tolist())
        docs.append(d)
    return docs


def pad(docs, field, S, width):
    out = np.zeros((len(docs), S, width), dtype=np.int32)
    lens = np.zeros((len(docs), S), dtype=np.int32)
    for e, d in enumerate(docs):
        for t in range(S):
            l = d[field][t]
            out[e, t, :len(l)] = l
            lens[e, t] = len(l)
    return out, lens


def main():
    sys.path.insert(0, REF)
    import graph_loader as gl
    rng = np.random.Generator(np.random.PCG64(4))
    blob = {}
    for tag, U, I, S, K, Fu, Fi in (("f34", 30, 40, 6, 4, 3, 4), ("f12", 25, 35, 5, 3, 1, 2)):
        start_time = 0
        user_docs = make_docs(rng, U, 1, U + 1, U + I + 1, S, 12, "uid")
        item_docs = make_docs(rng, I, U + 1, 1, U + 1, S, 12, "iid")
        base = 1 + U + I
        ufeat = {str(u): rng.integers(base, base + 9, Fu - 1).tolist() for u in range(1, U + 1)}
        ifeat = {str(i): rng.integers(base + 9, base + 40, Fi - 1).tolist() for i in range(U + 1, U + I + 1)}
        tmp = tempfile.mkdtemp()
        uf = itf = None
        if Fu > 1:
            uf = os.path.join(tmp, "u.pkl")
            pickle.dump(ufeat, open(uf, "wb"))
        if Fi > 1:
            itf = os.path.join(tmp, "i.pkl")
            pickle.dump(ifeat, open(itf, "wb"))
        h = gl.GraphHandler(S, "nodb", K, U, I, start_time, 10000, 10000, "rs", uf, itf, Fu, Fi)
        h.user_colls = [FakeColl(user_docs, "uid")]
        h.item_colls = [FakeColl(item_docs, "iid")]
        for pred_time in (S - 3, S - 1):
            np.random.seed(123)
            uids = rng.integers(1, U + 1, 5).tolist()
            iids = rng.integers(U + 1, U + I + 1, 5).tolist()
            u1, u2, i1, i2 = [], [], [], []
            for u in uids:
                a, b = h.gen_user_history(u, pred_time)
                u1.append(a)
                u2.append(b)
            for it in iids:
                a, b = h.gen_item_history(it, pred_time)
                i1.append(a)
                i2.append(b)
            p = "%s_p%d/" % (tag, pred_time)
            blob[p + "uids"], blob[p + "iids"] = np.asarray(uids), np.asarray(iids)
            blob[p + "user_1hop"] = np.asarray(u1).astype(np.int32)     # float 0.0 dummies -> 0
            blob[p + "user_2hop"] = np.asarray(u2).astype(np.int32)
            blob[p + "item_1hop"] = np.asarray(i1).astype(np.int32)
            blob[p + "item_2hop"] = np.asarray(i2).astype(np.int32)
        for nm, docs in (("user", user_docs), ("item", item_docs)):
            blob["%s/%s_1hop" % (tag, nm)], blob["%s/%s_1hop_len" % (tag, nm)] = pad(docs, "1hop", S, 9)
            blob["%s/%s_2hop" % (tag, nm)], blob["%s/%s_2hop_len" % (tag, nm)] = pad(docs, "2hop", S, 12)
        blob[tag + "/user_feat"] = np.asarray([[u] + (ufeat[str(u)] if Fu > 1 else []) for u in range(1, U + 1)], dtype=np.int32)
        blob[tag + "/item_feat"] = np.asarray([[i] + (ifeat[str(i)] if Fi > 1 else []) for i in range(U + 1, U + I + 1)], dtype=np.int32)
        blob[tag + "/dims"] = np.asarray([U, I, S, K, Fu, Fi, start_time])
        # mode 'is' (degree-weighted 2-hop draws, graph_loader.py:94-167): the degree lists + the reference's own draws
        # for a few entities, as counts per drawn id (many calls; np.random's stream)
        blob[tag + "/user_degrees"], _ = pad(user_docs, "degrees", S, 12)
        blob[tag + "/item_degrees"], _ = pad(item_docs, "degrees", S, 12)
        h_is = gl.GraphHandler(S, "nodb", K, U, I, start_time, 10000, 10000, "is", uf, itf, Fu, Fi)
        h_is.user_colls, h_is.item_colls = h.user_colls, h.item_colls
        np.random.seed(321)
        calls = 300
        for nm, ents, fn, first in (("user", [1, 2, 3], h_is.gen_user_history, U + 1), ("item", [U + 1, U + 2], h_is.gen_item_history, 1)):
            draws = np.zeros((len(ents), S - 1, calls * K), dtype=np.int32)
            for ei, e in enumerate(ents):
                for c in range(calls):
                    _, two = fn(e, S - 1)
                    for t in range(S - 1):
                        draws[ei, t, c * K:(c + 1) * K] = np.asarray(two[t], dtype=np.float64)[:, 0].astype(np.int32)
            blob["%s_is/%s_ents" % (tag, nm)] = np.asarray(ents)
            blob["%s_is/%s_draws" % (tag, nm)] = draws
    np.savez_compressed(os.path.join(HERE, "g4_loader.npz"), **blob)
    print("wrote g4_loader.npz", os.path.getsize(os.path.join(HERE, "g4_loader.npz")))


if __name__ == "__main__":
    main()
