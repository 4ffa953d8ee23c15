"""The three offline steps between a raw Tmall behaviour log and the hot path's inputs, as far as the
cfg-1 plumbing needs them (BASELINE.json configs[0]: bundled sample -> graph store -> loader -> train/eval).
Host-side NumPy/Python, run once per dataset exactly like the reference's scripts; nothing here is on the
step path.

  remap_tmall_log      code/feateng_tmall.py:30-133   one contiguous id space starting at 1: users, items,
                                                       cats, sellers, brands, ages, genders; feature rows
  tmall_slice_index    code/feateng_tmall.py:10-11,53-56   15-day time slices counted from 2015-05-01
  gen_target_lines     code/gen_target.py:99-121       'uid,pos_iid,neg...' lines of one prediction slice

The reference enumerates each id vocabulary through ``list(set(...))`` (feateng_tmall.py:58-64), i.e. in
Python's per-process string-hash order: its remap is not reproducible from run to run.  Here every
vocabulary is numbered in order of first appearance in the log -- a valid instance of the same layout.
``user_info_format1.csv`` (age, gender per user) is missing from the reference mount; callers synthesise
it (age = uid % 9, gender = uid % 3).
"""
import datetime

import numpy as np

TMALL_START = datetime.date(2015, 5, 1)     # feateng_tmall.py:10
TMALL_SLICE_DAYS = 15                       # feateng_tmall.py:11


def tmall_slice_index(mmdd):
    """time_stamp column 'MMDD' (as int) -> 15-day slice index (feateng_tmall.py:53-56)."""
    mmdd = np.asarray(mmdd).astype(np.int64)
    out = np.empty(mmdd.shape, dtype=np.int32)
    cache = {}
    for i, v in enumerate(mmdd.reshape(-1).tolist()):
        t = cache.get(v)
        if t is None:
            t = cache[v] = (datetime.date(2015, v // 100, v % 100) - TMALL_START).days // TMALL_SLICE_DAYS
        out.reshape(-1)[i] = t
    return out


def _first_appearance(col):
    """value -> 0-based rank of its first appearance"""
    uniq, first, inv = np.unique(col, return_index=True, return_inverse=True)
    rank = np.empty(len(uniq), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
    return rank[inv.reshape(-1)], len(uniq)


def remap_tmall_log(log, age=None, gender=None):
    """log: int array [n, >=6] with columns user_id, item_id, cat_id, seller_id, brand_id, time_stamp(MMDD)
    (tests/golden/tmall_sample_log.npz).  age / gender: per-row arrays (the joined user profile,
    feateng_tmall.py:13-28); default = the synthesised profile uid % 9 / uid % 3.

    Returns a dict: uid, iid (remapped ids per row), t_idx, n_users, n_items, feature_size (the reference
    prints it as 'feat size', :104), user_rows [U, 3] = [uid, age_id, gender_id] and item_rows [I, 4] =
    [iid, cat_id, seller_id, brand_id] (user_feat_dict / item_feat_dict, :125-133, keyed by row order)."""
    log = np.asarray(log)
    raw_u, raw_i, raw_c, raw_s, raw_b = (log[:, j].astype(np.int64) for j in range(5))
    if age is None:
        age = raw_u % 9
    if gender is None:
        gender = raw_u % 3
    cols = [raw_u, raw_i, raw_c, raw_s, raw_b, np.asarray(age).astype(np.int64), np.asarray(gender).astype(np.int64)]
    base = 1                                         # remap_id starts at 1; 0 is the dummy node (:74)
    ids, sizes = [], []
    for c in cols:                                   # users, items, cats, sellers, brands, ages, genders (:83-103)
        r, n = _first_appearance(c)
        ids.append((r + base).astype(np.int32))
        sizes.append(n)
        base += n
    uid, iid, cid, sid, bid, aid, gid = ids
    U, I = sizes[0], sizes[1]
    user_rows = np.zeros((U, 3), dtype=np.int32)
    item_rows = np.zeros((I, 4), dtype=np.int32)
    # later rows overwrite earlier ones, as the dict assignments of :130-131 do
    user_rows[uid - 1] = np.stack([uid, aid, gid], axis=1)
    item_rows[iid - 1 - U] = np.stack([iid, cid, sid, bid], axis=1)
    return dict(uid=uid, iid=iid, t_idx=tmall_slice_index(log[:, 5]), n_users=U, n_items=I, feature_size=int(base),
                user_rows=user_rows, item_rows=item_rows, vocab_sizes=tuple(sizes))


def gen_target_lines(uid, iid, t_idx, n_users, n_items, pred_time, neg_sample_num, start_time=0, seed=11):
    """TargetGen.gen_target_file (gen_target.py:99-121) on an in-memory log: one line per user that has at
    least one interaction in slice `pred_time` AND a history in [start_time, pred_time) (user_hist_dict,
    :225-236): (uid, [pos_iid, neg...]) with pos_iid the user's FIRST item of the slice in log order and the
    negatives uniform over the item id range (gen_user_neg_items with no pop list, :88-90).  Users are
    visited in id order, as the collection scan does."""
    rng = np.random.Generator(np.random.PCG64(seed))
    uid, iid, t_idx = np.asarray(uid), np.asarray(iid), np.asarray(t_idx)
    has_hist = np.zeros(n_users + 1, dtype=bool)
    has_hist[uid[(t_idx >= start_time) & (t_idx < pred_time)]] = True
    first_pos = {}
    for u, i, t in zip(uid.tolist(), iid.tolist(), t_idx.tolist()):
        if t == pred_time and u not in first_pos:
            first_pos[u] = i
    lines = []
    for u in sorted(first_pos):
        if has_hist[u]:
            negs = rng.integers(n_users + 1, n_users + n_items + 1, neg_sample_num).tolist()
            lines.append((u, [first_pos[u]] + negs))
    return lines


# constants of the Tmall run, code/score/train_score.py:45-54,339-364 and code/graph_storage.py:39-47
TMALL = dict(obj_per_time_slice=10, time_slice_num=12, graph_time_slice_num=14, start_time=0, user_fnum=3, item_fnum=4,
             pred_time_train=9, pred_time_validation=10, pred_time_test=11, eb_dim=16, hidden_size=32,
             eval_batch_size=100, train_neg=1, test_neg=99, max_1hop=10, max_2hop=100)


def tmall_pipeline(log, seed=11):
    """Raw Tmall log -> (TemporalGraph, remap dict, {'train' | 'validation' | 'test': target lines}).  Train
    lines carry 1 negative (train_score.py:18), validation / test lines 99 (:19)."""
    from .graph import TemporalGraph
    r = remap_tmall_log(log)
    c = TMALL
    g = TemporalGraph.from_log(r["uid"], r["iid"], r["t_idx"], r["n_users"], r["n_items"], c["graph_time_slice_num"],
                               r["user_rows"], r["item_rows"], max_1hop=c["max_1hop"], max_2hop=c["max_2hop"], seed=seed)
    targets = {}
    for name, pt, neg in (("train", c["pred_time_train"], c["train_neg"]),
                          ("validation", c["pred_time_validation"], c["test_neg"]),
                          ("test", c["pred_time_test"], c["test_neg"])):
        targets[name] = gen_target_lines(r["uid"], r["iid"], r["t_idx"], r["n_users"], r["n_items"], pt, neg,
                                         c["start_time"], seed + pt)
    return g, r, targets
