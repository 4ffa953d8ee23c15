"""Data-parallel SCoRe with the embedding table row-sharded across the GPUs of a node.

The reference is single-process, single-device (no NCCL/MPI anywhere), so
this layer has no counterpart to mirror; it implements what BASELINE.json's north_star
asks for: rows of emb_mtx live on shard ``row % G`` (range sharding would put every hot
categorical row on the last GPU -- id layout of feateng_tmall.py:72-101), each rank trains
on its own batch, and one training step exchanges

  1. int32 row requests       all_to_all_v   (unique rows of the batch, grouped by owner)
  2. fp32 rows                all_to_all_v   (owners gather from their shard)
  3. fp32 row gradients       all_to_all_v   (back to the owners, which sum them per row)
  4. dense-variable gradients all_reduce     (~0.5-11 MB)

over RCCL (``backend="nccl"`` on ROCm): all-to-all uses every xGMI link of a GPU at once.
Dense variables are replicated; Adam on the shard and on the replicas needs no further
communication.  The loss is the mean over the GLOBAL batch (G * B samples), so G ranks
with batch B reproduce one device with batch G*B (tests/test_dist_*.py).

Compute goes through a backend object: ``HipBackend`` (libscore_hip.so, the product path).
Tests inject a CPU backend to exercise this file's routing with gloo; nothing here falls
back to it on its own.
"""
import ctypes as C
import os
import sys
import threading

import numpy as np
import torch

# The sharded step uses five streams; HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues in first-use
# order and two streams on one queue serialise (1.89 - 2.30 ms/step from process to process with 4, 1.77 with 8).  The
# runtime reads the variable when it starts: this only helps when score_amd.dist is imported before the first HIP
# call of the process (bench.py sets it itself; INTEGRATION.md says so for other callers).
if not torch.cuda.is_initialized():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib
from .model import SCOREBASE, DeviceBatch, _ptr, ADAM_B1, ADAM_B2, ADAM_EPS


def _a2a_probe_hung(rank, world, timeout_s):
    """probe_a2a's watchdog: the list-form all-to-all (or the verdict's all-reduce behind it) has not returned"""
    sys.stderr.write("[score_amd.dist] rank %d of %d: the list-form all-to-all probe (empty own slots) has not returned after %.0f s; "
                     "start the job with SCORE_A2A=split (bench.py --a2a split) to use all_to_all_single only\n"
                     % (rank, world, timeout_s))
    sys.stderr.flush()
    os._exit(3)


def _to_host(t):
    """Device -> host copy that waits on the CURRENT stream only (pinned buffer + event), so a side
    stream's read-back does not stall behind work queued on other streams."""
    if t.device.type != "cuda":
        return t
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    ev.synchronize()
    return h


def _concurrent_stream(device, candidates=8, cycles=1500000):
    """A torch stream whose kernels really run beside the current stream's.  HIP deals streams onto a few hardware
    queues in first-use order, and with the model's, the communicators' and the caller's streams around, two of them
    can end up on the same queue -- where they serialise (seen in a kernel trace: the gradient-exchange chain ran
    BEHIND the weight-gradient tail it was meant to hide under).  So: try a few streams, time a spin kernel on each
    together with one on the current stream, take the first pair that overlaps.  ~1 ms per candidate, once."""
    main = torch.cuda.current_stream(device)
    mk = lambda: torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(candidates):
        st = torch.cuda.Stream(device=device)
        spans = []
        for rep in range(2):                      # (the first use of a stream sets its queue up: timed on the second)
            e0, e1, f1 = mk(), mk(), mk()
            torch.cuda.synchronize(device)
            e0.record(main)
            torch.cuda._sleep(cycles)
            e1.record(main)
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
                f1.record(st)
            torch.cuda.synchronize(device)
            spans.append((e0.elapsed_time(e1), e0.elapsed_time(f1)))
        single, span = spans[-1]
        if best is None:
            best = st
        if span < 1.4 * single:
            return st
    return best


def _a2a(cm, out, inp, out_splits, in_splits, tag):
    """the remote-only all-to-all where the communicator has one (TorchDistComm: tagged for bench.py's per-rank table)"""
    if isinstance(cm, TorchDistComm):
        return cm.all_to_all_remote(out, inp, out_splits, in_splits, tag=tag)
    return getattr(cm, "all_to_all_remote", cm.all_to_all)(out, inp, out_splits, in_splits)


class TorchDistComm(object):
    """torch.distributed collectives (nccl == RCCL on ROCm; gloo for the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._gloo = dist.get_backend(group) == "gloo"
        # all_to_all_remote's form: "remote" = the list form with EMPTY tensors in the own slot (what a rank owns never goes through
        # the collective), "split" = all_to_all_single with the own segment inside.  Decided ONCE, at set-up, identically on every
        # rank (probe_a2a): never by catching an error in the hot path, where a rank-local or asynchronous failure would flip one
        # rank only and the ranks' collectives would stop matching.  "split" unless SCORE_A2A=remote|probe (bench.py --a2a) asks otherwise.
        self._list_form = None
        self.a2a_probe = None   # what the probe found (bench.py prints it)
        self.last = None       # ("name", sequence number) of the collective this rank entered last (bench.py's heartbeat)
        self._seq = 0
        # bench.py --gpus N: a pair of timing events around every data-path collective of the timed steps, on the stream the
        # collective is issued on -- the per-rank table the first multi-GPU curve is read against
        self.timing = False
        self._timed = []

    def _timed_call(self, tag, tensor, fn):
        if not self.timing or tensor.device.type != "cuda":
            return fn()
        st = torch.cuda.current_stream(tensor.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        r = fn()
        e1.record(st)
        self._timed.append((tag, e0, e1, int(tensor.numel()) * tensor.element_size()))
        return r

    def timing_summary(self):
        """{tag: {"calls", "ms_avg", "ms_max", "bytes_avg"}} of the collectives timed so far (synchronises)"""
        out = {}
        for tag, e0, e1, nbytes in self._timed:
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            d = out.setdefault(tag, {"calls": 0, "ms_sum": 0.0, "ms_max": 0.0, "bytes_sum": 0})
            d["calls"] += 1; d["ms_sum"] += ms; d["ms_max"] = max(d["ms_max"], ms); d["bytes_sum"] += nbytes
        self._timed = []
        return {k: {"calls": v["calls"], "ms_avg": v["ms_sum"] / v["calls"], "ms_max": v["ms_max"],
                    "bytes_avg": v["bytes_sum"] / v["calls"]} for k, v in out.items()}

    def probe_a2a(self, device, force=None, timeout_s=None):
        """Settle on ONE all-to-all form for the whole run -- the same one on every rank -- by running it on tiny tensors and
        comparing what arrives with what every rank must receive (the verdicts are all-reduced).  One line on stderr from rank 0.
        A form nobody asked for is never entered (a hang of the list form on one rank must not be able to take a job down at
        set-up), so:
          not forced      "split" only: all_to_all_single, the plain RCCL path.  A rank's own segment rides through the collective
                          as a local copy -- 1 / G of a few MB per step, nothing a first multi-GPU run should risk a hang for.
          "remote"        (force or SCORE_A2A) ONLY the list form with empty own slots (a rank's own rows stay out of the
                          collective); a mismatch raises.
          "split"         the default, said explicitly.
          "probe"         "split" first, then the list form under a watchdog: if it has not returned after timeout_s
                          (SCORE_A2A_PROBE_TIMEOUT, default 60 s) the rank says so on stderr, names the switch that avoids it and
                          exits 3 (a collective that never completes cannot be abandoned inside the process); the list form is
                          taken when it round-trips on every rank, "split" otherwise."""
        force = force or os.environ.get("SCORE_A2A") or None
        if force not in (None, "split", "remote", "probe"):
            raise ValueError("all-to-all form must be 'split', 'remote' or 'probe', got %r" % (force,))
        asked, both = force, force == "probe"
        force = None if both else (force or "split")
        if self.world == 1 or self._gloo:
            self._list_form = None if self.world == 1 else True
            self.a2a_probe = {"form": "none (one rank)" if self.world == 1 else "pairwise (gloo)", "forced": force}
            return self.a2a_probe
        if timeout_s is None:
            timeout_s = float(os.environ.get("SCORE_A2A_PROBE_TIMEOUT", "60"))
        G, r = self.world, self.rank
        per = 3                                       # rows per peer; row values name (source, destination, row)
        inp = torch.tensor([[1000.0 * r + 10.0 * p + i for i in range(per)] for p in range(G)], dtype=torch.float32,
                           device=device).reshape(G * per, 1)
        want = torch.tensor([[1000.0 * p + 10.0 * r + i for i in range(per)] for p in range(G)], dtype=torch.float32,
                            device=device).reshape(G * per, 1)
        splits = [per] * G
        verdict = {}
        for form in (("split", "remote") if both else (force,)):
            ok = 1.0
            out = torch.full_like(inp, -1.0)
            dog = None
            if form == "remote" and both and timeout_s > 0:
                dog = threading.Timer(timeout_s, _a2a_probe_hung, args=(r, G, timeout_s))
                dog.daemon = True
                dog.start()
            try:
                if form == "split":
                    self.dist.all_to_all_single(out, inp, splits, splits, group=self.group)
                else:
                    outs, ins = list(out.split(splits, 0)), list(inp.split(splits, 0))
                    outs[r].copy_(ins[r])
                    outs[r] = out.new_empty((0, 1))
                    ins[r] = inp.new_empty((0, 1))
                    self.dist.all_to_all(outs, ins, group=self.group)
                if torch.device(device).type == "cuda":
                    torch.cuda.synchronize(device)
                ok = 1.0 if torch.equal(out, want) else 0.0
            except (RuntimeError, ValueError, TypeError) as e:      # (an argument check: before anything is enqueued)
                ok = 0.0
                verdict[form + "_error"] = str(e).splitlines()[0][:160]
            flag = torch.tensor([ok], dtype=torch.float32, device=device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)     # every rank, or nobody
            verdict[form] = bool(flag.item() == 1.0)
            if dog is not None:
                dog.cancel()
        if force is not None and not verdict[force]:
            raise RuntimeError("all-to-all form %r was asked for and does not round-trip on this backend: %r" % (force, verdict))
        form = force or ("remote" if verdict["remote"] else "split")
        if not verdict[form]:
            raise RuntimeError("neither all-to-all form round-trips on this backend: %r" % (verdict,))
        self._list_form = form == "remote"
        verdict.update(form=form, forced=asked)
        self.a2a_probe = verdict
        if r == 0:
            sys.stderr.write("[score_amd.dist] all-to-all probe over %d ranks: %r\n" % (G, verdict))
        return verdict

    def _note(self, name):
        self._seq += 1
        self.last = (name, self._seq)

    def index_comm(self):
        """The communicator of the index-only collectives (split sizes, row requests) and of the dense all-reduce: this
        communicator itself.  Every collective of a rank goes through ONE process group in host program order, which is
        the same on every rank, so no cross-rank ordering hazard exists by construction.  (Rounds 1 - 3 kept a second,
        opt-in communicator for the index traffic, SCORE_DUAL_COMM: two communicators in flight on different streams hang
        if two ranks' hardware queues ever serialise their kernels in opposite orders, it bought 2 % with one rank and could
        never be verified on more: removed.)"""
        return self

    def exchange_counts(self, send_counts, device, extra=None):
        """all_to_all of one count per peer.  `extra` (an int): sent to every peer beside its count -- the reply is
        (counts, extras), one extra per rank (the local batch sizes ride here, no second collective)."""
        if self.world == 1:            # nobody to tell: no collective, no read-back
            return list(send_counts) if extra is None else (list(send_counts), [int(extra)])
        if extra is None:
            t = torch.tensor(send_counts, dtype=torch.int64, device=device)
            out = torch.empty_like(t)
            self.all_to_all(out, t, [1] * self.world, [1] * self.world)
            return [int(x) for x in _to_host(out).tolist()]
        t = torch.tensor([[c, int(extra)] for c in send_counts], dtype=torch.int64, device=device)
        out = torch.empty_like(t)
        self.all_to_all(out, t, [1] * self.world, [1] * self.world)
        got = _to_host(out).tolist()
        return [int(r[0]) for r in got], [int(r[1]) for r in got]

    def all_to_all(self, out, inp, out_splits, in_splits):
        self._note("all_to_all[%s x %d]" % (str(inp.dtype).replace("torch.", ""), int(inp.shape[0]) if inp.dim() else 1))
        if not self._gloo:
            self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
            return
        # gloo has no all_to_all_v for every dtype/shape: pairwise isend/irecv.  Device tensors are staged through
        # host memory (gloo moves host buffers): that is how several processes rehearse the HIP path on ONE GPU
        # (tests/test_gpu_dist_procs.py); production multi-GPU runs use the RCCL branch above.
        outs = list(out.split(out_splits, 0))
        ins = list(inp.split(in_splits, 0))
        outs[self.rank].copy_(ins[self.rank])
        reqs = []
        for p in range(self.world):
            if p == self.rank:
                continue
            if in_splits[p] > 0:
                reqs.append(self.dist.isend(ins[p].contiguous().cpu(), p, group=self.group))
        for p in range(self.world):
            if p == self.rank or out_splits[p] == 0:
                continue
            buf = torch.empty(outs[p].shape, dtype=outs[p].dtype)
            self.dist.recv(buf, p, group=self.group)
            outs[p].copy_(buf)
        for r in reqs:
            r.wait()

    def all_to_all_remote(self, out, inp, out_splits, in_splits, tag="a2a"):
        return self._timed_call(tag, inp, lambda: self._all_to_all_remote(out, inp, out_splits, in_splits))

    def _all_to_all_remote(self, out, inp, out_splits, in_splits):
        """all_to_all of the segments that belong to OTHER ranks only; this rank's own segment is a device-side copy (or,
        for callers that pass out / inp whose own segments are the same memory, nothing).  What a rank owns never goes
        through RCCL: one rank = no collective at all, G ranks = (G - 1) / G of the bytes in the collective's kernels."""
        r = self.rank
        outs = list(out.split(out_splits, 0))
        ins = list(inp.split(in_splits, 0))
        if outs[r].numel() and outs[r].data_ptr() != ins[r].data_ptr():
            outs[r].copy_(ins[r])
        if self.world == 1:
            return
        if self._gloo:                 # (the pairwise host-staged exchange below already leaves the own segment out)
            self.all_to_all(out, inp, out_splits, in_splits)
            return
        self._note("all_to_all_remote[%s x %d]" % (str(inp.dtype).replace("torch.", ""), int(inp.shape[0]) - int(ins[r].shape[0])))
        if self._list_form is None:    # (nobody ran the set-up probe: ShardedSCORE does; a bare communicator settles it here)
            self.probe_a2a(inp.device)
        if not self._list_form:
            self.all_to_all(out, inp, out_splits, in_splits)
            return
        outs[r] = out.new_empty((0,) + tuple(out.shape[1:]))
        ins[r] = inp.new_empty((0,) + tuple(inp.shape[1:]))
        self.dist.all_to_all(outs, ins, group=self.group)

    def all_reduce_sum(self, t, tag="all_reduce"):
        return self._timed_call(tag, t, lambda: self._all_reduce_sum(t))

    def _all_reduce_sum(self, t):
        self._note("all_reduce[%d]" % t.numel())
        if self._gloo and t.device.type != "cpu":
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class _ShardModel(SCOREBASE):
    """SCOREBASE whose `table` is one row shard: local row i holds global row i*G + rank."""
    model_type = "SCORE"
    # (the time-tiled table optimizer works on a shard as on the whole table: the rows other ranks ask for are the
    #  "batch", HipBackend.gather catches them up before it reads them; on from 256 MB of sweep traffic per shard)
    # No device-side guard on a shard's optimizer (SCOREBASE._guard_on): a batch with an id outside the table is known to
    # every rank BEFORE its step starts (its index plan's status word rides with the row counts) and is rejected there
    _guard_on = False

    def __init__(self, rank, world, model_type, feature_size, *args, **kw):
        self.model_type = model_type
        self.rank, self.world = rank, world
        self.N_global = int(feature_size)
        self._rows_local = (self.N_global + world - 1) // world
        SCOREBASE.__init__(self, feature_size, *args, **kw)

    def _table_rows(self, feature_size):
        return self._rows_local

    def _init_params(self, seed):
        # shard-local: the initialiser is a pure function of (seed, global row, column), so this shard holds
        # exactly the rows the single-device model of the same seed holds -- and never materialises the others
        self._init_table(seed, self.world, self.rank, self.N_global)
        self._init_dense(seed)

    def set_params(self, params):
        """from the FULL variable set (emb_mtx [N, D]): keeps this shard's rows, sliced on the host"""
        emb = np.asarray(params["emb_mtx"], dtype=np.float32)
        self.row0 = emb[0].copy()
        part = np.ascontiguousarray(emb[self.rank::self.world])
        self.table.zero_()
        self.table[:part.shape[0]].copy_(torch.from_numpy(part))
        if self.rank == 0:
            self.table[0].zero_()                                    # global row 0: the masked dummy row
        self._set_dense(params)

    # checkpoint hooks: the file of a shard holds the shard's rows (local row 0 is the dummy row on rank 0 only)
    def _table_host(self):
        t = self.table.cpu().numpy()
        if self.rank == 0:
            t[0] = self.row0
        return t

    def _table_load(self, emb):
        emb = np.asarray(emb, dtype=np.float32)
        if emb.shape != tuple(self.table.shape):
            raise ValueError("emb_mtx shard shape %s != %s" % (emb.shape, tuple(self.table.shape)))
        self.table.copy_(torch.from_numpy(emb))
        if self.rank == 0:
            self.row0 = emb[0].copy()
            self.table[0].zero_()


class HipBackend(object):
    """Per-rank compute of the sharded step on libscore_hip.so."""

    def __init__(self, rank, world, model_type, cfg_args, seed=1111, device=None):
        self.m = _ShardModel(rank, world, model_type, *cfg_args, seed=seed, device=device)
        self.rank, self.world = rank, world
        self.device = self.m.device
        self.D = int(cfg_args[1])
        self.lib = self.m.lib
        self._scratch = None
        # time-tiled table optimizer or per-step sweep?  Decided once from the first steps' row requests (note_requests)
        self.auto_sweep = True
        self._req_seen, self._want_sweep = [], False
        self.defer_sweep = False         # gather(): leave the optimizer's window slice to the next backward pass
        # one id-status word per plan slot (score_state_t.id_status of THAT plan only: plans run a step ahead of the
        # compute, a shared sticky word would blame the step in flight for the next batch's ids)
        self._plan_status = torch.zeros((4,), dtype=torch.int32, device=self.device)

    # -- index plan ---------------------------------------------------------------------
    def plan_launch(self, batch_data, slot=0):
        """Enqueue the index plan on the current stream and start the read-back of its sizes; no host wait."""
        m = self.m
        db = m.device_batch(batch_data)
        lay, ws = m._workspace(db.B, slot)
        st = m._state(ws)
        word = self._plan_status[slot % 4:slot % 4 + 1]
        word.zero_()
        st.id_status = _ptr(word)
        _lib.check(self.lib.score_index_plan(C.byref(m.cfg), C.byref(st), C.byref(db.struct), self.world, 1,
                                             m._stream()), "score_index_plan")
        meta = ws[lay.plan_meta:lay.plan_meta + 2 + self.world].view(torch.int32)
        host = torch.empty((meta.numel() + 1,), dtype=meta.dtype, pin_memory=True)
        host[:meta.numel()].copy_(meta, non_blocking=True)
        host[meta.numel():].copy_(word, non_blocking=True)        # (the plan's occurrence fill saw every id as fed)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return dict(db=db, lay=lay, ws=ws, host=host, event=ev, slot=slot)

    def plan_finish(self, h):
        """Wait for the plan's sizes (that event only) and build the plan record."""
        h["event"].synchronize()
        db, lay, ws = h["db"], h["lay"], h["ws"]
        meta = h["host"].tolist()
        U, offs, bad_ids = meta[0], meta[1:2 + self.world], meta[2 + self.world]
        uniq = ws[lay.plan_unique_rows:lay.plan_unique_rows + U].view(torch.int32)
        # batch struct over the remapped (unique-position) index tensors
        sizes = [t.numel() for t in (db.tensors[0], db.tensors[3], db.tensors[1], db.tensors[2], db.tensors[4],
                                     db.tensors[5])]
        rm = [ws[lay.plan_remap[g]:lay.plan_remap[g] + sizes[g]].view(torch.int32) for g in range(6)]
        # plan order: user_1hop, item_2hop, user_2hop, item_1hop, target_user, target_item
        remapped = _lib.Batch(_ptr(rm[0]), _ptr(rm[2]), _ptr(rm[3]), _ptr(rm[1]), _ptr(rm[4]), _ptr(rm[5]),
                              _ptr(db.tensors[6]), _ptr(db.tensors[7]), db.B, getattr(db, "active_slices", 0))
        # the record keeps (lay, ws): the step that consumes the plan must use THIS buffer -- the sort output, the
        # unique-row list and the remapped indices live in it -- whatever the workspace cache does in between
        return dict(db=db, remapped=remapped, keep=rm, U=U, offsets=offs, unique_rows=uniq, slot=h["slot"],
                    event=h["event"], lay=lay, ws=ws, bad_ids=bad_ids & 63)

    def plan(self, batch_data, slot=0):
        return self.plan_finish(self.plan_launch(batch_data, slot))

    def gather(self, req_rows):
        m = self.m
        n = req_rows.numel()
        out = torch.empty((n, self.D), dtype=torch.float32, device=self.device)
        if m._tiled_on():
            # time-tiled optimizer: the requested rows (local indices) up to date first, then this step's slice of
            # the shard, on this same stream (the shard path runs enough streams already)
            # The window slice: in a training step it is left pending and started by the backward pass at its stage
            # boundary 2, beside the recurrence (ShardedSCORE.forward_backward) -- inline here it sat on the
            # gradient-exchange chain the main stream waits for at the end of the step (0.13 ms of ~0.4); on a stream
            # of its own right here it measured 2.25-2.29 ms/step against 1.76.
            m._catchup_ids([req_rows] if n else [], True, inline_sweep=not self.defer_sweep)
        else:
            m._flush_adam()
        if n:
            _lib.check(self.lib.score_gather_fwd(_ptr(m._tbl), m._tbl.shape[0], self.D, _ptr(req_rows),
                                                 n, _ptr(out), m._stream()), "score_gather_fwd")
        return out

    def _state(self, plan, mini):
        m = self.m
        lay, ws = plan["lay"], plan["ws"]
        return lay, ws, _lib.State(table=_ptr(mini), n_table_rows=mini.shape[0], w=_ptr(m.w), workspace=_ptr(ws),
                                   workspace_bytes=ws.numel() * 4, scatter_mode=2, global_batch=int(m.global_batch),
                                   gemm_mode=int(m.gemm_mode), debug_flags=int(m.debug_flags), context=m._ctx,
                                   id_status=None)     # (the ids were checked by this batch's score_index_plan: plan["bad_ids"])

    def forward(self, plan, mini, reg_lambda, keep_prob, masks):
        m = self.m
        lay, ws, st = self._state(plan, mini)
        m0 = m1 = None
        if masks is not None:
            m0 = torch.as_tensor(np.asarray(masks[0]), dtype=torch.uint8).to(self.device).contiguous()
            m1 = torch.as_tensor(np.asarray(masks[1]), dtype=torch.uint8).to(self.device).contiguous()
        seed = (m._drop_seed * 0x9E3779B1 + m.step * 0x85EBCA77 + self.rank * 0xC2B2AE35) & 0xFFFFFFFFFFFFFFFF
        rc = self.lib.score_forward(C.byref(m.cfg), C.byref(st), C.byref(plan["remapped"]), float(reg_lambda),
                                    float(keep_prob), _ptr(m0), _ptr(m1), C.c_uint64(seed),
                                    m._event_array(m.fwd_events), m._stream())
        _lib.check(rc, "score_forward")
        self._keep = (m0, m1)
        B = plan["db"].B
        return dict(lay=lay, ws=ws, st=st, y_pred=ws[lay.y_pred:lay.y_pred + B], loss=ws[lay.loss:lay.loss + 3])

    def backward(self, plan, mini, fw, keep_prob, scatter_event=None):
        """scatter_event (a torch.cuda.Event already recorded once): recorded by score_backward on its stream as
        soon as the row gradients are complete (stage boundary 4), i.e. before the weight-gradient products of
        the pass -- the caller starts the row-gradient exchange behind it.  Returns (mini_g, that event)."""
        m = self.m
        # every unique position holds a row with at least one use in the batch, so the pull scatter stores it
        # exactly once; position 0 (the dummy row, whose uses are skipped) is the only one left to clear
        mini_g = torch.empty_like(mini)
        mini_g[0].zero_()
        events = list(m.bwd_events) if m.bwd_events else None
        if scatter_event is not None:
            if events is None:
                events = [None] * 6
            if events[4] is None:
                events[4] = scatter_event
        self.sweep_start_event = None
        if m._pending_sweep is not None:       # time-tiled optimizer: the window slice starts at stage boundary 2
            if m._ev_stage is None:
                m._ev_stage = torch.cuda.Event()
                m._ev_stage.record(torch.cuda.current_stream(self.device))     # materialise the hipEvent_t
            if events is None:
                events = [None] * 6
            if events[2] is None:
                events[2] = m._ev_stage
            self.sweep_start_event = events[2]
        rc = self.lib.score_backward(C.byref(m.cfg), C.byref(fw["st"]), C.byref(plan["remapped"]), float(keep_prob),
                                     _ptr(m.w_g), _ptr(mini_g), m._event_array(events), m._stream())
        _lib.check(rc, "score_backward")
        if scatter_event is not None:
            return mini_g, events[4]
        return mini_g

    def launch_sweep(self, stream=None):
        """the pending window slice of the shard's optimizer: behind backward's stage boundary 2 on `stream` (the
        caller's side stream), or on the current stream"""
        m = self.m
        if m._pending_sweep is None:
            return
        cur = torch.cuda.current_stream(self.device)
        if stream is None or self.sweep_start_event is None:
            m._launch_sweep(cur)
            return
        stream.wait_event(self.sweep_start_event)
        with torch.cuda.stream(stream):
            m._launch_sweep(stream)

    def dense_grad_with_loss(self, fw):
        """[n_w + 4] buffer: the dense gradient followed by this rank's share of the global log-loss mean
        (so one all-reduce carries both); element n_w is the global log-loss afterwards."""
        m = self.m
        m._w_g_ext[m.n_w:m.n_w + 1].copy_(fw["loss"][1:2])
        return m._w_g_ext

    def accumulate(self, req_rows, grads_in, counts=None):
        """Combine the row gradients received from every rank into this shard's table gradient.
        counts[p] = rows rank p asked for (its slice of req_rows, unique inside the slice): one
        score_rows_accumulate per rank, in rank order -- no sort, no atomics, reproducible."""
        m = self.m
        # no zero fill: rows are marked (state 2) as they are written and score_adam_rows reads
        # gradient rows in that state only
        m._begin_row_grads()
        if counts is None:
            counts = [req_rows.numel()]
        if len(counts) > 1 and len(counts) <= 64:
            # every source rank's list in one launch (the lists are unique and ascending: segments of the plans' unique-row
            # lists); the same bits as one launch per source in rank order, which at eight ranks were eight small launches
            # in a row on the chain the next step's rows wait for
            offs = (C.c_int64 * (len(counts) + 1))(*np.concatenate([[0], np.cumsum([int(c) for c in counts])]).tolist())
            rc = self.lib.score_rows_accumulate_multi(_ptr(req_rows), _ptr(grads_in), offs, len(counts), self.D,
                                                      m._tbl.shape[0], _ptr(m.table_g), _ptr(m.table_flags), m._stream())
            _lib.check(rc, "score_rows_accumulate_multi")
            return
        off = 0
        for c in counts:
            c = int(c)
            if c:
                rc = self.lib.score_rows_accumulate(_ptr(req_rows[off:off + c]), _ptr(grads_in[off:off + c]), c,
                                                    self.D, m._tbl.shape[0], _ptr(m.table_g), _ptr(m.table_flags),
                                                    m._stream())
                _lib.check(rc, "score_rows_accumulate")
            off += c

    def dense_grad(self):
        return self.m.w_g

    def adam(self, lr, reg_lambda):
        self.m.apply_adam(lr, reg_lambda)

    # the two halves of the update, for the pipelined step: the shard's rows need the row gradients only
    def note_requests(self, n_req):
        """n_req: rows all ranks together asked this shard for in one step.  The time-tiled optimizer saves the
        traffic of the rows that get NO gradient in a step; with many ranks (weak scaling: the global batch grows with
        G while the shard shrinks) most of a shard gets one every step -- cfg-3: 12 % of the rows at one rank, 23 %
        at two, 46 % at four, 92 % at eight -- and the per-step sweep is the cheaper one.  Decided once, from the mean
        of the first four steps; applied at the next optimizer call (the stream the table's work is ordered on)."""
        if self._req_seen is None or not self.auto_sweep:
            return
        self._req_seen.append(int(n_req))
        if len(self._req_seen) >= 4:
            frac = float(np.mean(self._req_seen)) / max(1, self.m._tbl.shape[0])
            self._want_sweep = frac > 0.35
            self._req_seen = None

    def adam_table(self, lr):
        m = self.m
        if self._want_sweep:
            self._want_sweep = False
            m.adam_window = 0          # (the setter applies what is owed first)
        if m._tiled_on() and m._row_grads:
            m._adam_table_tiled(lr)
        else:
            m.adam_table(lr)

    def adam_dense(self, lr, reg_lambda):
        self.m.adam_dense(lr, reg_lambda)
        self.m.adam_advance()

    def set_global_batch(self, n):
        self.m.global_batch = int(n)

    def labels(self, plan):
        return plan["db"].tensors[6]


class ShardedSCORE(object):
    """SCORE(...) with the reference's train/eval signatures (score.py:101-133), table sharded
    over the ranks of `comm`.  Every rank calls train()/eval() with its own batch_data."""

    model_type = "SCORE"

    def __init__(self, feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice, user_fnum, item_fnum,
                 seed=1111, comm=None, backend=None, model_type=None, device=None):
        self.comm = comm if comm is not None else TorchDistComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        if model_type is not None:
            self.model_type = model_type
        cfg_args = (feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice, user_fnum, item_fnum)
        self.backend = backend if backend is not None else HipBackend(self.rank, self.world, self.model_type,
                                                                      cfg_args, seed, device)
        self.device = self.backend.device
        self.D = int(eb_dim)
        self._side, self._slot, self._slot_done, self._prefetched = None, 0, [None, None, None], None
        self._gside = None
        self._ready = None           # (batch, plan, mini-table) fetched for the next step by the pipelined one
        self._adam_done = False
        if self.device.type == "cuda" and hasattr(self.comm, "index_comm"):
            # bring the communicator(s) up now (every rank constructs the model): their lazy first-use
            # initialisation costs tens of ms and would otherwise land inside a training step
            for cm in {id(c): c for c in (self.comm, self.comm.index_comm())}.values():
                cm.exchange_counts([0] * self.world, self.device)
        if hasattr(self.comm, "probe_a2a"):
            # which all-to-all form this backend round-trips: settled here, once, the same on every rank
            self.comm.probe_a2a(self.device)
        if self.device.type == "cuda" and hasattr(self.backend, "dense_grad_with_loss"):
            # the gradient-exchange stream (and its start-up probe, a handful of device-wide waits): here, not
            # inside the first training step
            self._gside = _concurrent_stream(self.device)
            self._ev_scatter = torch.cuda.Event()
            self._ev_scatter.record(torch.cuda.current_stream(self.device))      # materialise the hipEvent_t

    # bench.py compatibility with the single-device model
    @property
    def n_w(self):
        return self.backend.m.n_w

    def device_batch(self, batch_data):
        return self.backend.m.device_batch(batch_data) if hasattr(self.backend, "m") else batch_data

    def feed(self, batches, depth=2):
        """SCOREBASE.feed: the host conversion of the next feed tuples on a worker thread, one or two batches ahead"""
        m = getattr(self.backend, "m", None)
        return m.feed(batches, depth) if m is not None else iter(batches)

    def enable_stage_events(self, on=True):
        self.backend.m.enable_stage_events(on)

    fwd_events = property(lambda self: self.backend.m.fwd_events,
                          lambda self, v: setattr(self.backend.m, "fwd_events", v))
    bwd_events = property(lambda self: self.backend.m.bwd_events,
                          lambda self, v: setattr(self.backend.m, "bwd_events", v))

    # -- step phases -----------------------------------------------------------------------
    def _request(self, plan, cm=None):
        """Tell every owner which of its rows this rank needs (two small collectives)."""
        cm = cm if cm is not None else self.comm
        offs = plan["offsets"]
        send = [offs[o + 1] - offs[o] for o in range(self.world)]      # unique rows I need from shard o
        B_local = plan["B"] if "B" in plan else plan["db"].B
        # rows shard-me must serve to rank p; every rank's batch size rides along: the loss is the mean over the
        # GLOBAL batch (sum of the local ones -- per-rank loaders end with short last batches of different sizes)
        # ... and so does the status word of the batch's index plan (six bits, one per id tensor of the feed tuple): every
        # rank knows which ranks fed an id outside the table before anybody starts the step (_reject_bad_ids)
        recv, extras = cm.exchange_counts(send, self.device, extra=int(B_local) | (int(plan.get("bad_ids", 0)) << 32))
        sizes = [e & 0xFFFFFFFF for e in extras]
        if self.world == 1:
            req = plan["unique_rows"]           # (the one owner is this rank: its request list IS the plan's unique rows)
        else:
            req = torch.empty((sum(recv),), dtype=torch.int32, device=self.device)
            _a2a(cm, req, plan["unique_rows"], recv, send, "a2a_row_requests")
        plan.update(send=send, recv=recv, req=req, global_B=sum(sizes), bad_by_rank=[e >> 32 for e in extras])
        if hasattr(self.backend, "note_requests"):
            self.backend.note_requests(sum(recv))
        return plan

    def _plan_and_request(self, batch_data, slot=0, cm=None):
        """Index-only phase (needs no parameters): plan the batch, then request its rows."""
        be = self.backend
        plan = be.plan(batch_data, slot) if slot else be.plan(batch_data)
        return self._request(plan, cm)

    def _rows(self, plan):
        """Parameter phase: owners gather the requested rows from their (up-to-date) shard."""
        be, cm = self.backend, self.comm
        rows = be.gather(plan["req"])
        if self.world == 1:
            return rows                         # (gathered in the plan's unique-row order: it IS the mini-table)
        mini = torch.empty((plan["U"], self.D), dtype=torch.float32, device=self.device)
        _a2a(cm, mini, rows, plan["send"], plan["recv"], "a2a_rows")
        return mini

    def prefetch(self, batch_data):
        """Index-only phase of the NEXT batch in one call (plan, then request its rows).
        forward_backward(..., next_batch=) only LAUNCHES the plan (before this step's compute is enqueued)
        and lets the next step pick it up.  Every rank must call it, with its own next batch."""
        self._prefetch_launch(batch_data)
        self._prefetch_finish()

    def _prefetch_launch(self, batch_data):
        """Start the next batch's index plan on a high-priority side stream; returns at once.  Called
        BEFORE this step's forward/backward are enqueued, so the plan's kernels run under them.  Three
        workspace slots keep the plan buffers of the steps in flight apart."""
        be = self.backend
        if self.device.type != "cuda" or not hasattr(be, "plan_launch"):
            self._prefetched = (batch_data, None, None, None)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device, priority=-1)
        slot = (self._slot + 1) % 3
        if self._slot_done[slot] is not None:                # the step that last used this slot (t-2)
            self._side.wait_event(self._slot_done[slot])
        with torch.cuda.stream(self._side):
            handle = be.plan_launch(batch_data, slot)
        self._prefetched = (batch_data, handle, None, None)

    def _prefetch_finish(self):
        """Read the prefetched plan's sizes and request its rows, from the side stream and on the index
        communicator (TorchDistComm.index_comm: not queued behind the previous step's gradient exchange).
        Runs at the start of the step that uses the plan: it was launched a whole step earlier, so the two
        host reads return at once while the GPU still has that previous step's tail in its queue."""
        batch_data, handle, _, _ = self._prefetched
        if handle is None:                                   # CPU test backend: nothing to overlap
            self._prefetched = (batch_data, None, self._plan_and_request(batch_data, 0), None)
            return
        icm = self.comm.index_comm() if hasattr(self.comm, "index_comm") else self.comm
        with torch.cuda.stream(self._side):
            plan = self._request(self.backend.plan_finish(handle), icm)
            ev = self._side.record_event()
        self._prefetched = (batch_data, None, plan, ev)

    def _fetch(self, batch_data):
        """plan -> request rows from their owners -> gathered [U, D] mini-table"""
        rd = getattr(self, "_ready", None)
        self._ready = None
        if rd is not None and rd[0] is batch_data:      # fetched by the previous (pipelined) step
            self._slot = rd[1].get("slot", 0)
            self._prefetched = None
            return rd[1], rd[2]
        pf = getattr(self, "_prefetched", None)
        if pf is not None and pf[0] is batch_data:
            if pf[2] is None:
                self._prefetch_finish()
                pf = self._prefetched
            plan = pf[2]
            if pf[3] is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(pf[3])
                plan["req"].record_stream(cur)          # allocated on the side stream, consumed here
            self._slot = plan.get("slot", 0)
        else:
            plan = self._plan_and_request(batch_data, self._slot if self.device.type == "cuda" else 0)
        self._prefetched = None
        return plan, self._rows(plan)

    def _reject_bad_ids(self, plan):
        """ValueError on EVERY rank, before anything of the step has run, if any rank's batch holds a feature id outside
        [0, feature_size): tf.nn.embedding_lookup raises inside sess.run and no variable is updated (score.py:51-66,
        101-116).  The index plan of each rank's batch reported its tensors (score_state_t.id_status of that plan), the
        words were exchanged with the row counts: no extra collective, no device read-back, the same decision on every
        rank.  Rows fetched ahead for the rejected batch are dropped; the next call plans afresh."""
        bad = plan.get("bad_by_rank")
        if not bad or not any(bad):
            return
        self._ready = self._prefetched = None
        from .model import BATCH_FIELDS
        msgs = ["rank %d: %s" % (r, ", ".join("batch_data[%d] (%s)" % (i, BATCH_FIELDS[i]) for i in range(6) if b >> i & 1))
                for r, b in enumerate(bad) if b]
        m = getattr(self.backend, "m", None)
        raise ValueError("feature id outside [0, %s) -- %s (tf.nn.embedding_lookup would raise: score.py:51-66); the batch "
                         "was rejected before its step started: no variable was updated on any rank"
                         % (m.N_global if m is not None else "feature_size", "; ".join(msgs)))

    def _mark_step_end(self):
        if self.device.type == "cuda":
            self._slot_done[self._slot] = torch.cuda.current_stream(self.device).record_event()

    def forward_backward(self, batch_data, reg_lambda, keep_prob=1.0, dropout_masks=None, next_batch=None, lr=None):
        """lr given (with next_batch, HIP backend): the PIPELINED step -- the optimizer runs inside.  The shard's
        table update needs only the row gradients, which exist before the weight-gradient products that end the
        backward pass; so behind the scatter a side stream runs: row-gradient all-to-all -> owner-side accumulate ->
        ApplyAdam over the shard -> gather of the rows the NEXT batch asked for -> their all-to-all, all under this
        step's weight-gradient tail, the dense all-reduce (on the index communicator, so it is not queued behind the
        row traffic) and the dense ApplyAdam.  The next step starts with its mini-table in hand.  Returns with
        the update applied; do not call apply_adam after it."""
        be, cm = self.backend, self.comm
        if hasattr(be, "defer_sweep"):
            be.defer_sweep = self.device.type == "cuda"
        plan, mini = self._fetch(batch_data)
        self._reject_bad_ids(plan)
        if next_batch is not None:
            self._prefetch_launch(next_batch)     # its kernels run under this step's forward
        be.set_global_batch(plan["global_B"])         # sum of every rank's batch size (exchanged with the row counts)
        fw = be.forward(plan, mini, reg_lambda, keep_prob, dropout_masks)
        if self.device.type == "cuda" and hasattr(be, "dense_grad_with_loss"):
            # The row gradients are complete before the weight-gradient products of the backward pass
            # (score_backward's stage boundary 4): their all-to-all and the owner-side accumulate start there, on a
            # side stream, under those products.  The dense gradient goes out at the end of the pass in ONE
            # all-reduce together with the log-loss share.
            cur = torch.cuda.current_stream(self.device)
            mini_g, ev = be.backward(plan, mini, fw, keep_prob, scatter_event=self._ev_scatter)
            be.launch_sweep(self._gside)      # (first on that stream: it is through before the row gradients are)
            # (one rank: the gradient rows of the mini-table ARE the owner's input, in request order: no exchange, no copy)
            grads_in = mini_g if self.world == 1 else \
                torch.empty((plan["req"].numel(), self.D), dtype=torch.float32, device=self.device)
            pipelined = lr is not None and next_batch is not None and self._prefetched is not None \
                and self._prefetched[1] is not None
            nxt = None
            if pipelined:       # the next batch's plan was launched a forward + backward ago: sizes and requests now
                self._prefetch_finish()
                nxt = self._prefetched
                self._prefetched = None
            icm = self.comm.index_comm() if hasattr(self.comm, "index_comm") else cm
            self._gside.wait_event(ev)
            with torch.cuda.stream(self._gside):
                if self.world > 1:
                    _a2a(cm, grads_in, mini_g, plan["recv"], plan["send"], "a2a_row_grads")
                mini_g.record_stream(self._gside)
                be.accumulate(plan["req"], grads_in, plan["recv"])
                if pipelined:
                    be.adam_table(lr)
            # Host order = execution order on a communicator, identical on every rank: row gradients, then the dense
            # all-reduce (on the critical path: the dense ApplyAdam and the next forward wait for it), then the next
            # batch's rows (their all-to-all overlaps the dense ApplyAdam)
            buf = be.dense_grad_with_loss(fw)
            (icm if pipelined else cm).all_reduce_sum(buf)
            if pipelined:
                be.adam_dense(lr, reg_lambda)
            with torch.cuda.stream(self._gside):
                if pipelined:
                    plan_n = nxt[2]
                    self._gside.wait_event(nxt[3])          # its row requests have arrived
                    plan_n["req"].record_stream(self._gside)
                    mini_n = self._rows(plan_n)
                    self._ready = (next_batch, plan_n, mini_n)
                done = self._gside.record_event()
            cur.wait_event(done)            # (also orders the frees of mini_g / grads_in behind their last use)
            if pipelined:
                mini_n.record_stream(cur)
                self._mark_step_end()
            self._adam_done = pipelined
            n_w = buf.numel() - 4
            return (None, buf[n_w], fw["loss"][2]), fw      # [-, global log-loss, l2]
        mini_g = be.backward(plan, mini, fw, keep_prob)
        if hasattr(be, "launch_sweep"):
            be.launch_sweep(None)
        if self.world == 1:
            grads_in = mini_g
        else:
            grads_in = torch.empty((plan["req"].numel(), self.D), dtype=torch.float32, device=self.device)
            _a2a(cm, grads_in, mini_g, plan["recv"], plan["send"], "a2a_row_grads")
        cm.all_reduce_sum(be.dense_grad())
        be.accumulate(plan["req"], grads_in, plan["recv"])
        loss = fw["loss"].clone()          # [loss, log_loss (local share of the global mean), l2]
        cm.all_reduce_sum(loss[1:2])
        return loss, fw

    def apply_adam(self, lr, reg_lambda):
        if getattr(self, "_adam_done", False):      # the pipelined forward_backward has applied it already
            self._adam_done = False
            return
        self.backend.adam(lr, reg_lambda)
        self._mark_step_end()

    def train_async(self, batch_data, lr, reg_lambda, keep_prob=0.8, dropout_masks=None, next_batch=None):
        loss, _ = self.forward_backward(batch_data, reg_lambda, keep_prob, dropout_masks, next_batch, lr=lr)
        self.apply_adam(lr, reg_lambda)
        return loss[1] + float(reg_lambda) * loss[2]

    def train(self, sess, batch_data, lr, reg_lambda, keep_prob=0.8, dropout_masks=None, next_batch=None):
        return float(self.train_async(batch_data, lr, reg_lambda, keep_prob, dropout_masks, next_batch).item())

    def check_ids(self, collective=True):
        """Kept for callers of train_async / eval_async written against SCOREBASE.check_ids: nothing can be pending
        here.  A batch with a feature id outside the table is rejected when its step (or eval) STARTS -- ValueError on
        every rank, before any kernel of it has run (_reject_bad_ids) -- so no later sync point has anything to report."""
        return None

    # -- checkpoint (score.py:135-142), one file per rank ---------------------------------------
    def _shard_path(self, path):
        return "%s.shard%d-of-%d" % (path, self.rank, self.world)

    def save(self, sess, path):
        """Every rank writes its own file `<path>.shard<r>-of-<G>.npz`: its row shard of emb_mtx with both Adam
        slots, plus the (replicated) dense variables and their slots under the TF variable names -- the
        single-device checkpoint format (SCOREBASE.save) with emb_mtx holding rows r, r+G, r+2G, ...  Call it on
        every rank; no collective is involved."""
        m = self.backend.m
        torch.cuda.current_stream(self.device).synchronize()
        if self._gside is not None:
            self._gside.synchronize()
        m.save(sess, self._shard_path(path))

    def restore(self, sess, path):
        """Inverse of save(); the world size must be the one the checkpoint was written with."""
        m = self.backend.m
        f = self._shard_path(path)
        if not os.path.exists(f + ".npz"):
            raise FileNotFoundError("%s.npz: no shard file for rank %d of %d (checkpoints are per world size)" %
                                    (f, self.rank, self.world))
        self._ready = self._prefetched = None        # rows fetched ahead belong to the old parameters
        m.restore(sess, f)

    def eval(self, sess, batch_data, reg_lambda):
        be = self.backend
        if hasattr(be, "defer_sweep"):
            be.defer_sweep = False
        plan, mini = self._fetch(batch_data)
        self._reject_bad_ids(plan)
        B = plan["B"] if "B" in plan else plan["db"].B
        be.set_global_batch(B)             # eval reports the local batch's loss, as the reference does
        fw = be.forward(plan, mini, reg_lambda, 1.0, None)
        self._mark_step_end()
        pred = fw["y_pred"].cpu().numpy().reshape([-1, ]).tolist()
        label = be.labels(plan).cpu().numpy().reshape([-1, ]).tolist()
        loss = fw["loss"]
        val = float((loss[1] + float(reg_lambda) * loss[2]).item())
        return pred, label, val
