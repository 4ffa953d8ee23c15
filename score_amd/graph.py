"""In-memory temporal graph store + device-side batch loader.

Replaces, for the training loop, the reference's MongoDB documents (code/graph_storage.py:78-246:
per entity and time slice a 1-hop list, a sampled 2-hop list) and its loader processes
(code/score/graph_loader.py:279-402).  The graph is built once on the host (it is preprocessing in the
reference too), lives in HBM as CSR, and every batch is assembled by one HIP launch
(score_batch_assemble, include/score_hip.h) into the int32 tensors score.py:21-30 feeds on.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .model import DeviceBatch, _ptr, carve_batch, flat_batch_size


def _csr(lists_per_cell):
    lens = np.fromiter((len(l) for l in lists_per_cell), dtype=np.int64, count=len(lists_per_cell))
    off = np.zeros(len(lists_per_cell) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    nbr = np.fromiter((x for l in lists_per_cell for x in l), dtype=np.int32, count=int(off[-1]))
    return off, nbr


class TemporalGraph(object):
    """CSR over (entity, slice) for both sides + feature rows.  Ids: users 1..U, items U+1..U+I
    (feateng_tmall.py:72-101)."""

    def __init__(self, n_users, n_items, time_slice_num, user_csr, item_csr, user_rows, item_rows):
        self.U, self.I, self.S = int(n_users), int(n_items), int(time_slice_num)
        self.user_csr, self.item_csr = user_csr, item_csr          # dicts: off1, nbr1, off2, nbr2
        self.user_rows = np.ascontiguousarray(user_rows, dtype=np.int32)
        self.item_rows = np.ascontiguousarray(item_rows, dtype=np.int32)
        assert self.user_rows.shape[0] == self.U and self.item_rows.shape[0] == self.I
        self.Fu, self.Fi = self.user_rows.shape[1], self.item_rows.shape[1]
        self._dev = None
        self.user_degrees = self.item_degrees = None      # mode 'is': degree behind every 2-hop entry (aligned with nbr2)

    # ---- construction -----------------------------------------------------------------
    @classmethod
    def from_padded(cls, n_users, n_items, S, u1, u1len, u2, u2len, i1, i1len, i2, i2len, user_rows, item_rows,
                    user_deg=None, item_deg=None):
        """From padded per-(entity, slice) lists: x[e, t, :xlen[e, t]].  user_deg / item_deg: the 'degrees' lists
        padded like u2 / i2 (mode 'is')."""
        def side(a1, l1, a2, l2):
            c1 = [a1[e, t, :l1[e, t]].tolist() for e in range(a1.shape[0]) for t in range(S)]
            c2 = [a2[e, t, :l2[e, t]].tolist() for e in range(a2.shape[0]) for t in range(S)]
            o1, n1 = _csr(c1)
            o2, n2 = _csr(c2)
            return dict(off1=o1, nbr1=n1, off2=o2, nbr2=n2)
        g = cls(n_users, n_items, S, side(u1, u1len, u2, u2len), side(i1, i1len, i2, i2len), user_rows, item_rows)
        if user_deg is not None:
            flat = lambda a, l: _csr([a[e, t, :l[e, t]].tolist() for e in range(a.shape[0]) for t in range(S)])[1]
            g.user_degrees, g.item_degrees = flat(user_deg, u2len), flat(item_deg, i2len)
        return g

    @classmethod
    def from_log(cls, uid, iid, t_idx, n_users, n_items, time_slice_num, user_rows, item_rows,
                 max_1hop=10, max_2hop=100, seed=11):
        """From a remapped behaviour log, as GraphStore does (graph_storage.py:93-128 construct_coll_1hop,
        :130-246 construct_coll_2hop; pinned by tests/golden/g5_graph_store.npz, documents written by the
        reference itself):
          1-hop of (node, slice t) = its neighbours in log order, repeats kept (:118-121);
          2-hop = for each of the node's neighbours (all of them, or a random max_1hop of them when there
            are more: the reference SHUFFLES the stored 1-hop list in place and keeps the first max_1hop,
            :166-168, so the stored list of such a cell ends up permuted too) whose own slice-t degree is > 1,
            that neighbour's slice-t list cut to its first max_1hop entries (:171-176), concatenated; more than
            max_2hop entries are down-sampled without replacement (:180-183); `degrees` holds, aligned with
            it, the degree of the neighbour each entry came through (what mode 'is' weights by);
          items first, then users (:147-189, :193-237): the user pass cuts item lists the item pass has
            already permuted.
        The random choices come from `seed`, not from the reference's `random` / `np.random` streams."""
        rng = np.random.Generator(np.random.PCG64(seed))
        S, U, I = int(time_slice_num), int(n_users), int(n_items)
        uid, iid, t_idx = (np.asarray(a).astype(np.int64) for a in (uid, iid, t_idx))
        # Round 3: sort-based CSR instead of one Python list per (entity, slice) cell (21 M lists at Tmall scale).  The
        # result is the list build's, element for element (tests/test_graph_store.py compares the two and the
        # reference's own documents): a stable sort keeps the log order inside a cell, and the random choices are drawn
        # in the list build's order -- cell by cell, a cell's 1-hop shuffle before its 2-hop down-sampling -- by a loop
        # over the few cells that need one.

        def one_hop(ent0, nbr, n_ent):
            key = ent0 * S + t_idx
            order = np.argsort(key, kind="stable")
            off = np.zeros(n_ent * S + 1, dtype=np.int64)
            np.cumsum(np.bincount(key, minlength=n_ent * S), out=off[1:])
            return off, nbr[order].astype(np.int32)
        uo1, un1 = one_hop(uid - 1, iid, U)
        io1, in1 = one_hop(iid - U - 1, uid, I)

        def expand(own_off, own_nbr, other_off, other_nbr, other_base, cells):
            """2-hop candidates of `cells` (sorted cell indices), before down-sampling: (values, degrees, lengths)"""
            len1 = np.minimum(own_off[cells + 1] - own_off[cells], max_1hop)        # the first max_1hop neighbours
            tot1 = int(len1.sum())
            cell_rep = np.repeat(np.arange(len(cells)), len1)
            pos = np.arange(tot1) - np.repeat(np.cumsum(len1) - len1, len1)
            x = own_nbr[own_off[cells][cell_rep] + pos].astype(np.int64)
            oc = (x - other_base) * S + (cells[cell_rep] % S)
            d = other_off[oc + 1] - other_off[oc]
            L = np.where(d > 1, np.minimum(d, max_1hop), 0)
            tot2 = int(L.sum())
            within = np.arange(tot2) - np.repeat(np.cumsum(L) - L, L)
            vals = other_nbr[np.repeat(other_off[oc], L) + within]
            degs = np.repeat(d, L).astype(np.int32)
            lens = np.bincount(np.repeat(cell_rep, L), minlength=len(cells)).astype(np.int64)
            return vals, degs, lens

        def two_hop(own_off, own_nbr, other_off, other_nbr, other_base, n_ent):
            ncell = n_ent * S
            all_cells = np.arange(ncell, dtype=np.int64)
            len1 = own_off[1:] - own_off[:-1]
            over1 = len1 > max_1hop
            # candidate counts of the cells whose neighbour set is already final (no 1-hop shuffle)
            _, _, lens = expand(own_off, own_nbr, other_off, other_nbr, other_base, all_cells)
            special = np.nonzero(over1 | (lens > max_2hop))[0]
            picks = {}
            for c in special.tolist():                     # the list build's order of random draws
                if over1[c]:
                    seg = own_nbr[own_off[c]:own_off[c + 1]]
                    seg[:] = seg[rng.permutation(len(seg))]                  # in place, like random.shuffle
                    lens[c] = expand(own_off, own_nbr, other_off, other_nbr, other_base, np.asarray([c]))[2][0]
                if lens[c] > max_2hop:
                    picks[c] = rng.permutation(int(lens[c]))[:max_2hop]
            vals, degs, lens = expand(own_off, own_nbr, other_off, other_nbr, other_base, all_cells)
            off = np.zeros(ncell + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            if picks:
                new_len = lens.copy()
                for c in picks:
                    new_len[c] = max_2hop
                noff = np.zeros(ncell + 1, dtype=np.int64)
                np.cumsum(new_len, out=noff[1:])
                src = np.arange(int(off[-1]), dtype=np.int64)
                cell_of = np.repeat(all_cells, lens)
                keep = np.ones(len(src), dtype=bool)
                for c in picks:
                    keep[off[c]:off[c + 1]] = False
                dst_idx = np.empty(int(noff[-1]), dtype=np.int64)
                kept = src[keep]
                dst_idx[noff[cell_of[keep]] + (kept - off[cell_of[keep]])] = kept
                for c, idx in picks.items():
                    dst_idx[noff[c]:noff[c + 1]] = off[c] + idx
                vals, degs, off = vals[dst_idx], degs[dst_idx], noff
            return off, vals.astype(np.int32), degs.astype(np.int32)
        io2, in2, ideg = two_hop(io1, in1, uo1, un1, 1, I)              # items first: the user pass below cuts item lists
        uo2, un2, udeg = two_hop(uo1, un1, io1, in1, U + 1, U)          # the item pass has already permuted
        g = cls(U, I, S, dict(off1=uo1, nbr1=un1, off2=uo2, nbr2=un2),
                dict(off1=io1, nbr1=in1, off2=io2, nbr2=in2), user_rows, item_rows)
        g.user_degrees, g.item_degrees = udeg, ideg                      # aligned with nbr2 (same offsets)
        return g

    @classmethod
    def _from_log_lists(cls, uid, iid, t_idx, n_users, n_items, time_slice_num, user_rows, item_rows,
                        max_1hop=10, max_2hop=100, seed=11):
        """The per-cell Python-list form of from_log (rounds 1 - 2), kept as the statement the sort-based build is
        tested against: same arguments, same result, element for element."""
        rng = np.random.Generator(np.random.PCG64(seed))
        S, U, I = time_slice_num, n_users, n_items
        u1 = [[] for _ in range(U * S)]
        i1 = [[] for _ in range(I * S)]
        for u, i, t in zip(np.asarray(uid).tolist(), np.asarray(iid).tolist(), np.asarray(t_idx).tolist()):
            u1[(u - 1) * S + t].append(i)
            i1[(i - U - 1) * S + t].append(u)

        def two_hop(own, other, other_base, n):
            out, deg = [], []
            for e in range(n):
                for t in range(S):
                    nb = own[e * S + t]
                    if len(nb) > max_1hop:
                        nb[:] = [nb[j] for j in rng.permutation(len(nb))]      # in place, like random.shuffle
                        nb = nb[:max_1hop]
                    acc, dg = [], []
                    for x in nb:
                        lst = other[(x - other_base) * S + t]
                        d = len(lst)
                        if d > 1:
                            acc += lst[:max_1hop]
                            dg += [d] * min(d, max_1hop)
                    if len(acc) > max_2hop:
                        idx = rng.permutation(len(acc))[:max_2hop]
                        acc = [acc[j] for j in idx]
                        dg = [dg[j] for j in idx]
                    out.append(acc)
                    deg.append(dg)
            return out, deg
        i2, ideg = two_hop(i1, u1, 1, I)
        u2, udeg = two_hop(u1, i1, U + 1, U)
        uo1, un1 = _csr(u1)
        uo2, un2 = _csr(u2)
        io1, in1 = _csr(i1)
        io2, in2 = _csr(i2)
        g = cls(U, I, S, dict(off1=uo1, nbr1=un1, off2=uo2, nbr2=un2),
                dict(off1=io1, nbr1=in1, off2=io2, nbr2=in2), user_rows, item_rows)
        g.user_degrees, g.item_degrees = _csr(udeg)[1], _csr(ideg)[1]      # aligned with nbr2 (same offsets)
        return g

    def cell(self, side, hop, e, t):
        """Neighbour list of 0-based entity e in slice t ('user' / 'item', hop 1 / 2) -- host view for tests."""
        c = self.user_csr if side == "user" else self.item_csr
        off, nbr = c["off%d" % hop], c["nbr%d" % hop]
        return nbr[off[e * self.S + t]:off[e * self.S + t + 1]]

    # ---- one file per graph (what a db_name of the reference resolves to, see GraphLoader) ---------------
    def save(self, path):
        """The graph as ONE .npz file: the CSR arrays of both sides, the feature rows, the 2-hop degree lists if kept."""
        a = {"dims": np.asarray([self.U, self.I, self.S], dtype=np.int64), "user_rows": self.user_rows, "item_rows": self.item_rows}
        for side, c in (("u", self.user_csr), ("i", self.item_csr)):
            for k in ("off1", "nbr1", "off2", "nbr2"):
                a["%s_%s" % (side, k)] = np.asarray(c[k])
        if self.user_degrees is not None and self.item_degrees is not None:
            a["u_deg2"], a["i_deg2"] = np.asarray(self.user_degrees, dtype=np.int32), np.asarray(self.item_degrees, dtype=np.int32)
        with open(path, "wb") as f:
            np.savez(f, **a)
        return path

    @classmethod
    def load(cls, path):
        with np.load(path) as z:
            U, I, S = [int(x) for x in z["dims"]]
            side = lambda p: {k: z["%s_%s" % (p, k)] for k in ("off1", "nbr1", "off2", "nbr2")}
            g = cls(U, I, S, side("u"), side("i"), z["user_rows"], z["item_rows"])
            if "u_deg2" in z.files:
                g.user_degrees, g.item_degrees = z["u_deg2"], z["i_deg2"]
        return g

    # ---- device residency -------------------------------------------------------------
    def to_device(self, device=None, mode="rs"):
        """mode: GraphHandler's 2-hop sampling mode -- 'rs' uniform (what train_score.py uses for all three data sets),
        'is' degree-weighted (graph_loader.py:94-167; needs the degree lists from_log / from_padded keep)."""
        if mode not in ("rs", "is"):
            raise ValueError("WRONG GRAPH_HANDLER MODE: {}".format(mode))          # graph_loader.py:248
        if mode == "is" and (self.user_degrees is None or self.item_degrees is None):
            raise ValueError("mode 'is' needs the 2-hop degree lists")
        if not torch.cuda.is_available():
            raise RuntimeError("TemporalGraph.to_device needs an AMD GPU (HIP); batch assembly has no CPU path")
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = {"u_" + k: t(v) for k, v in self.user_csr.items()}
        d.update({"i_" + k: t(v) for k, v in self.item_csr.items()})
        # an empty neighbour array still needs a valid pointer
        for k in list(d):
            if d[k].numel() == 0:
                d[k] = torch.zeros((1,), dtype=d[k].dtype, device=dev)
        d["user_rows"], d["item_rows"] = t(self.user_rows), t(self.item_rows)
        if mode == "is":
            for k, a in (("u_deg2", self.user_degrees), ("i_deg2", self.item_degrees)):
                d[k] = t(np.asarray(a, dtype=np.int32)) if len(a) else torch.zeros((1,), dtype=torch.int32, device=dev)
        self._dev = d
        self.device = dev
        self.mode = mode
        self.struct = _lib.Graph(_ptr(d["u_off1"]), _ptr(d["u_nbr1"]), _ptr(d["u_off2"]), _ptr(d["u_nbr2"]),
                                 _ptr(d["i_off1"]), _ptr(d["i_nbr1"]), _ptr(d["i_off2"]), _ptr(d["i_nbr2"]),
                                 _ptr(d["user_rows"]), _ptr(d["item_rows"]), self.U, self.I, self.S, self.Fu, self.Fi,
                                 1 if mode == "is" else 0, _ptr(d.get("u_deg2")), _ptr(d.get("i_deg2")))
        return self


class _AssembledBatch(DeviceBatch):
    """a batch assembled on the device (views of one flat buffer, like every DeviceBatch); indexable like the
    reference's 8-tuple"""

    def __init__(self, tensors, B, active=0, flat=None):
        self.tensors = tensors
        self.flat = flat
        self.B = B
        self.active_slices = int(active)  # every sample has length pred_time - start_time (graph_loader.py:382)
        self.struct = _lib.Batch(*[_ptr(t) for t in tensors], B, self.active_slices)

    def __len__(self):
        return 8

    def __getitem__(self, i):             # train_score.py:157 reads batch_data[5]
        return self.tensors[i]


class DeviceGraphLoader(object):
    """Iterator over device batches accepted by SCORE.train / SCORE.eval, from a TemporalGraph object and target lines in
    memory (`target_lines`: iterable of 'uid,pos_iid,neg...' strings or of (uid, [iids]) pairs -- gen_target.py's
    target_<t>.txt format).  NOT the reference's constructor: `GraphLoader` below has that one (graph_loader.py:279-281)
    and resolves its arguments to this class."""

    def __init__(self, graph, batch_size, target_lines, start_time, pred_time, neg_sample_num,
                 max_time_len, obj_per_time_slice, seed=1111):
        if batch_size % (1 + neg_sample_num) != 0:
            raise ValueError("batch size should be time of {}".format(1 + neg_sample_num))    # :289-291
        if graph._dev is None:
            graph.to_device()
        self.g, self.lib = graph, _lib.load()
        self.lines_per_batch = batch_size // (1 + neg_sample_num)
        self.neg, self.start_time, self.pred_time = neg_sample_num, start_time, pred_time
        self.T, self.K, self.seed = max_time_len, obj_per_time_slice, seed
        uids, iids = [], []
        for line in target_lines:
            if isinstance(line, str):
                parts = line.strip().split(',')
                u, its = int(parts[0]), [int(x) for x in parts[1:2 + neg_sample_num]]
            else:
                u, its = int(line[0]), [int(x) for x in line[1]][:1 + neg_sample_num]
            uids.append(u)
            iids += its
        self.uids = torch.tensor(uids, dtype=torch.int32, device=graph.device)
        self.iids = torch.tensor(iids, dtype=torch.int32, device=graph.device)
        self.n_lines = len(uids)
        self._pos = 0
        self._batch_no = 0

    def __iter__(self):
        return self

    def __len__(self):
        return (self.n_lines + self.lines_per_batch - 1) // self.lines_per_batch

    def __next__(self):
        if self._pos >= self.n_lines:
            raise StopIteration
        n = min(self.lines_per_batch, self.n_lines - self._pos)      # the last batch is short (:321-324)
        c = 1 + self.neg
        B, g, T, K = n * c, self.g, self.T, self.K
        shapes = ((B, T, K, g.Fi), (B, T, K, g.Fu), (B, T, K, g.Fu), (B, T, K, g.Fi), (B, g.Fu), (B, g.Fi), (B,), (B,))
        flat = torch.empty((flat_batch_size(shapes),), dtype=torch.int32, device=g.device)
        tens = carve_batch(flat, shapes)
        out = _lib.BatchOut(*[_ptr(t) for t in tens])
        u = self.uids[self._pos:self._pos + n]
        it = self.iids[self._pos * c:(self._pos + n) * c]
        rc = self.lib.score_batch_assemble(C.byref(g.struct), _ptr(u), _ptr(it), n, self.neg, T, K, self.start_time,
                                           self.pred_time, C.c_uint64(self.seed + 7919 * self._batch_no),
                                           C.byref(out), C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
        _lib.check(rc, "score_batch_assemble")
        self._pos += n
        self._batch_no += 1
        length = self.pred_time - self.start_time
        return _AssembledBatch(tens, B, length if 0 < length < T else 0, flat)


# ---- the reference's call site (train_score.py:150, 222) ---------------------------------------------------------------
_GRAPHS = {}


def register_graph(db_name, graph):
    """What `db_name` (graph_handler_params[1]: the MongoDB database graph_storage.py filled, e.g. 'tmall_2hop') stands for
    here: a TemporalGraph, or the path of a file written by TemporalGraph.save (read on first use).  The documents'
    content -- per (entity, slice) a 1-hop list and a sampled 2-hop list, graph_storage.py:78-246 -- is what
    TemporalGraph.from_log builds; the database itself is out of scope (storage engine)."""
    _GRAPHS[str(db_name)] = graph


def resolve_graph(db_name):
    g = _GRAPHS.get(str(db_name))
    if g is None:
        raise KeyError("no graph registered for db_name %r (score_amd.graph.register_graph(db_name, TemporalGraph | file))" % (db_name,))
    if not isinstance(g, TemporalGraph):
        g = TemporalGraph.load(g)
        _GRAPHS[str(db_name)] = g
    return g


class GraphLoader(DeviceGraphLoader):
    """GraphLoader(graph_handler_params, batch_size, target_file, start_time, pred_time, worker_n, neg_sample_num)
    -- the reference's constructor, argument for argument (graph_loader.py:279-281; graph_handler_params as GraphHandler
    takes them, :41-55: [time_slice_num, db_name, obj_per_time_slice, user_num, item_num, start_time, user_per_collection,
    item_per_collection, mode, user_feat_dict_file, item_feat_dict_file, user_fnum, item_fnum]), so train_score.py:150, 222
    swap by import as the model does.  Yields the same 8 tensors per batch (:383), on the device.

    db_name -> a registered TemporalGraph (register_graph); target_file is read as the reference reads it (:293-294);
    history length = time_slice_num - start_time - 1 (:252-254).  worker_n / max_q_size / wait_time are accepted and unused:
    a batch is ONE kernel launch, there are no loader processes to size.  user_per_collection / item_per_collection (the
    sharding of the Mongo collections) and the two feat-dict files (the graph carries the feature rows) are unused too.
    Errors: ValueError where the reference prints and exits (batch size not a multiple of 1 + neg_sample_num, :289-291; a
    mode other than 'is' / 'rs', :248) or where the graph contradicts the parameters."""

    def __init__(self, graph_handler_params, batch_size, target_file, start_time, pred_time, worker_n, neg_sample_num,
                 max_q_size=10, wait_time=0.01, seed=1111):
        (time_slice_num, db_name, obj_per_time_slice, user_num, item_num, gh_start_time, _upc, _ipc, mode, _uf, _if,
         user_fnum, item_fnum) = graph_handler_params
        g = resolve_graph(db_name)
        if (g.U, g.I, g.Fu, g.Fi) != (int(user_num), int(item_num), int(user_fnum), int(item_fnum)):
            raise ValueError("graph %r has (users, items, user_fnum, item_fnum) = %r, graph_handler_params say %r"
                             % (db_name, (g.U, g.I, g.Fu, g.Fi), (user_num, item_num, user_fnum, item_fnum)))
        if int(gh_start_time) != int(start_time):
            raise ValueError("GraphHandler start_time %r != GraphLoader start_time %r (one value in train_score.py)"
                             % (gh_start_time, start_time))
        if int(time_slice_num) > g.S or int(pred_time) > g.S:
            raise ValueError("the graph holds %d time slices, asked for %r (pred_time %r)" % (g.S, time_slice_num, pred_time))
        if g._dev is None or getattr(g, "mode", None) != mode:
            g.to_device(mode=mode)
        with open(target_file, "r") as f:
            lines = f.readlines()
        self.worker_n, self.max_q_size, self.wait_time = worker_n, max_q_size, wait_time
        super(GraphLoader, self).__init__(g, batch_size, lines, int(start_time), int(pred_time), int(neg_sample_num),
                                          int(time_slice_num) - int(gh_start_time) - 1, int(obj_per_time_slice), seed=seed)
        self.num_of_batch = len(self)           # (:295-297)

    def stop(self):
        """graph_loader.py:399-402 ends its processes here; nothing to end"""
        return None
