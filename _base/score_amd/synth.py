"""Deterministic synthetic Tmall-shaped batches.

Emits the 8-tuple GraphLoader yields (graph_loader.py:383) -- user_1hop
[B,T,K,Fi], user_2hop [B,T,K,Fu], item_1hop [B,T,K,Fu], item_2hop [B,T,K,Fi],
target_user [B,Fu], target_item [B,Fi], label [B], length [B] -- with the value
distribution the loader produces: one shared id space (feateng_tmall.py:72-101:
0 dummy, users, items, then categorical vocabularies), fixed per-entity side
features, cyclic padding of short 1-hop lists (graph_loader.py:181-182), 2-hop
draws with replacement (:192), all-zero dummy slices (:90-91), tail slices
replicating the last real slice (:254-256), user-side tensors shared by a
user's positive and negative candidate (:363-364).
"""
import numpy as np

# categorical vocab sizes per config: item side then user side
TMALL_ITEM_VOCABS = (1700, 5000, 8399)     # cat, seller, brand
TMALL_USER_VOCABS = (9, 3)                 # age, gender


def _zipf_cdf(n, s):
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    c = np.cumsum(w)
    return c / c[-1]


class SynthWorld(object):
    def __init__(self, n_users, n_items, T, K, user_fnum=3, item_fnum=4,
                 item_vocabs=TMALL_ITEM_VOCABS, user_vocabs=TMALL_USER_VOCABS, seed=1111):
        assert len(item_vocabs) >= item_fnum - 1 and len(user_vocabs) >= user_fnum - 1
        self.U, self.I, self.T, self.K = n_users, n_items, T, K
        self.Fu, self.Fi = user_fnum, item_fnum
        self.seed = seed
        rng = np.random.Generator(np.random.PCG64(seed))
        base = 1 + n_users + n_items
        self.item_feat = np.zeros((n_items, item_fnum - 1), dtype=np.int32)
        for j in range(item_fnum - 1):
            v = item_vocabs[j]
            self.item_feat[:, j] = base + np.searchsorted(_zipf_cdf(v, 1.0), rng.random(n_items))
            base += v
        self.user_feat = np.zeros((n_users, user_fnum - 1), dtype=np.int32)
        for j in range(user_fnum - 1):
            v = user_vocabs[j]
            self.user_feat[:, j] = base + np.searchsorted(_zipf_cdf(v, 1.0), rng.random(n_users))
            base += v
        self.feature_size = base
        self._ucdf = _zipf_cdf(n_users, 0.8)
        self._icdf = _zipf_cdf(n_items, 0.8)

    # entity index (0-based within type) -> feature row [id, side features...]
    def _user_rows(self, u0):
        u0 = np.asarray(u0)
        return np.concatenate([(u0 + 1)[..., None].astype(np.int32), self.user_feat[u0]], axis=-1)

    def _item_rows(self, i0):
        i0 = np.asarray(i0)
        return np.concatenate([(i0 + 1 + self.U)[..., None].astype(np.int32), self.item_feat[i0]], axis=-1)

    def _history(self, rng, n_ent, hop1_is_item, length):
        """[n_ent, T, K] 0-based neighbour picks for 1-hop and 2-hop, -1 = dummy."""
        T, K = self.T, self.K
        cdf1 = self._icdf if hop1_is_item else self._ucdf
        cdf2 = self._ucdf if hop1_is_item else self._icdf
        h1 = np.full((n_ent, T, K), -1, dtype=np.int64)
        h2 = np.full((n_ent, T, K), -1, dtype=np.int64)
        empty = rng.random((n_ent, length)) < 0.3
        deg = np.minimum(K, rng.geometric(0.15, (n_ent, length)))
        pool = rng.integers(1, 101, (n_ent, length))
        for e in range(n_ent):
            for t in range(length):
                if empty[e, t]:
                    continue
                d = int(deg[e, t])
                nb = np.unique(np.searchsorted(cdf1, rng.random(d)))
                rng.shuffle(nb)
                h1[e, t] = nb[np.arange(K) % len(nb)]            # cyclic pad
                cand = np.searchsorted(cdf2, rng.random(int(pool[e, t])))
                h2[e, t] = cand[rng.integers(0, len(cand), K)]   # with replacement
        if length > 0:
            h1[:, length:] = h1[:, length - 1:length]
            h2[:, length:] = h2[:, length - 1:length]
        return h1, h2

    def _expand(self, picks, as_item):
        """[.., K] 0-based picks (-1 dummy) -> [.., K, F] feature ids (0 for dummy)."""
        safe = np.maximum(picks, 0)
        rows = self._item_rows(safe) if as_item else self._user_rows(safe)
        rows = rows.copy()
        rows[picks < 0] = 0
        return rows.astype(np.int32)

    def batch(self, B, batch_idx=0, length=None, as_lists=False, per_user=2):
        """per_user = candidates per target line: 2 for training (one positive + one negative,
        graph_loader.py:289-292), 100 for the ranking evaluation (1 + 99, train_score.py:19)."""
        assert B % per_user == 0, "batches hold whole target lines (one positive + the negatives of a user)"
        rng = np.random.Generator(np.random.PCG64([self.seed, 7919, batch_idx]))
        T = self.T
        length = max(T - 2, 1) if length is None else length
        nu = B // per_user
        users = rng.integers(0, self.U, nu)
        items = rng.integers(0, self.I, B)
        u1, u2 = self._history(rng, nu, True, length)
        i1, i2 = self._history(rng, B, False, length)
        user_1hop = np.repeat(self._expand(u1, True), per_user, axis=0)
        user_2hop = np.repeat(self._expand(u2, False), per_user, axis=0)
        item_1hop = self._expand(i1, False)
        item_2hop = self._expand(i2, True)
        target_user = np.repeat(self._user_rows(users), per_user, axis=0).astype(np.int32)
        target_item = self._item_rows(items).astype(np.int32)
        label = (np.arange(B) % per_user == 0).astype(np.int32)
        length_arr = np.full((B,), length, dtype=np.int32)
        out = (user_1hop, user_2hop, item_1hop, item_2hop, target_user, target_item, label, length_arr)
        if as_lists:
            out = tuple(a.tolist() for a in out)
        return out


CONFIGS = {
    # name: (U, I, T, K, D, H, B, Fu, Fi, item_vocabs, user_vocabs)
    "tiny": (40, 60, 3, 2, 4, 8, 4, 3, 4, (5, 7, 6), (4, 3)),
    "cfg2": (100000, 50000, 10, 5, 16, 32, 256, 3, 4, (1500, 5000, 5000), (9, 3)),
    "tmall_default": (424170, 1090390, 11, 10, 16, 32, 200, 3, 4, TMALL_ITEM_VOCABS, TMALL_USER_VOCABS),
    "cfg3": (424170, 1090390, 20, 10, 64, 128, 1024, 3, 4, TMALL_ITEM_VOCABS, TMALL_USER_VOCABS),
    # the reference's other two data sets at its own hyper-parameters (train_score.py:15-16, 23-43, 285-338, 372):
    # CCMR  N = 1 + 4,920,695 + 190,129 + 80,172 + 213,482 + 63 + 1,044 = 5,405,586, T = 41 - 0 - 1 = 40 (train length 38),
    # Fu = 1, Fi = 5;  Taobao  N = 1 + 984,080 + 4,049,268 + 9,405 = 5,042,754, T = 8 (train length 6), Fu = 1, Fi = 2
    "ccmr_default": (4920695, 190129, 40, 10, 16, 32, 200, 1, 5, (80172, 213482, 63, 1044), ()),
    "taobao_default": (984080, 4049268, 8, 10, 16, 32, 200, 1, 2, (9405,), ()),
    "cfg5_taobao": (984080, 4049268, 50, 20, 128, 256, 4096, 1, 2, (9405,), ()),
    "cfg5_tmall": (984080, 4049268, 50, 20, 128, 256, 4096, 3, 4, (3000, 3000, 3396), (6, 3)),
}


def make_world(name, seed=1111):
    U, I, T, K, D, H, B, Fu, Fi, iv, uv = CONFIGS[name]
    w = SynthWorld(U, I, T, K, Fu, Fi, iv, uv, seed)
    return w, dict(feature_size=w.feature_size, eb_dim=D, hidden_size=H, max_time_len=T,
                   obj_per_time_slice=K, user_fnum=Fu, item_fnum=Fi, batch=B)


def lowdup_batch(n_rows, B, T, K, Fu, Fi, seed=0, length=None):
    """LOW-DUPLICATION probe batch for the gather kernels: every index uniform over [1, n_rows), no dummy
    slices, nothing shared between the candidates of a user, no replicated tail slices.  With n_rows far above
    the R * B row uses of the batch nearly every use is a distinct row, so the kernel's algorithmic bytes
     and its memory-side traffic coincide -- the case the HBM roofline is about.  Not a model of
    what the loader produces (SynthWorld.batch is)."""
    rng = np.random.Generator(np.random.PCG64([seed, 424242]))
    u = lambda *s: rng.integers(1, n_rows, s, dtype=np.int64).astype(np.int32)
    length = T if length is None else length
    return (u(B, T, K, Fi), u(B, T, K, Fu), u(B, T, K, Fu), u(B, T, K, Fi), u(B, Fu), u(B, Fi),
            (np.arange(B) % 2 == 0).astype(np.int32), np.full((B,), length, dtype=np.int32))


def make_graph(world, time_slice_num, seed=7, p_empty=0.3, max_1hop=10, max_2hop=16, active_users=None,
               active_items=None):
    """A synthetic TemporalGraph over `world`'s id space for loader-inclusive measurements: per (entity, slice)
    an empty cell with probability p_empty, else min(max_1hop, Geometric(0.15)) 1-hop neighbours (skewed over the
    opposite id range: rank ~ n * u^3) and a 2-hop pool of up to max_2hop.  Only the first active_users /
    active_items entities get histories (target lines must stay inside them); neighbours and feature rows span
    the whole id space.  Vectorised."""
    from .graph import TemporalGraph
    rng = np.random.Generator(np.random.PCG64([seed, 99]))
    S = time_slice_num

    def skewed(n, count):
        u = rng.random(count, dtype=np.float32)
        return np.minimum((n * u * u * u).astype(np.int64), n - 1)

    def side(n_ent, n_active, n1_range, base1, n2_range, base2):
        cells = n_ent * S
        act = n_active * S
        d1 = np.zeros(cells, dtype=np.int64)
        d1[:act] = np.minimum(max_1hop, rng.geometric(0.15, act))
        d1[:act][rng.random(act, dtype=np.float32) < p_empty] = 0
        d2 = np.zeros(cells, dtype=np.int64)
        d2[:act] = np.where(d1[:act] > 0, rng.integers(1, max_2hop + 1, act), 0)
        o1 = np.zeros(cells + 1, dtype=np.int64)
        o2 = np.zeros(cells + 1, dtype=np.int64)
        np.cumsum(d1, out=o1[1:])
        np.cumsum(d2, out=o2[1:])
        n1 = (skewed(n1_range, int(o1[-1])) + base1).astype(np.int32)
        n2 = (skewed(n2_range, int(o2[-1])) + base2).astype(np.int32)
        return dict(off1=o1, nbr1=n1, off2=o2, nbr2=n2)
    w = world
    au = w.U if active_users is None else min(active_users, w.U)
    ai = w.I if active_items is None else min(active_items, w.I)
    users = side(w.U, au, w.I, 1 + w.U, w.U, 1)
    items = side(w.I, ai, w.U, 1, w.I, 1 + w.U)
    g = TemporalGraph(w.U, w.I, S, users, items, w._user_rows(np.arange(w.U)), w._item_rows(np.arange(w.I)))
    g.active_users, g.active_items = au, ai
    return g
