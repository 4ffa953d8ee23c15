"""Drop-in host side of the SCoRe hot path on MI355X.

Mirrors the reference's model interface (code/score/score.py):
    SCORE(feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice, user_fnum, item_fnum)
    loss = model.train(sess, batch_data, lr, reg_lambda)            # score.py:101-116
    pred, label, loss = model.eval(sess, batch_data, reg_lambda)    # score.py:118-133
    model.save(sess, path) / model.restore(sess, path)              # score.py:135-142
plus the ablations RIA, RCA, SCORE_USER, SCORE_ITEM (score.py:227-369).  ``sess`` is
accepted and ignored (there is no TF session).  ``batch_data`` is the 8-tuple the
reference's GraphLoader yields (graph_loader.py:383): nested lists, ndarrays or
tensors, indexed positionally.

All compute runs in libscore_hip.so (hand-written HIP for gfx950) through the C-ABI of
include/score_hip.h; PyTorch only owns device memory and streams.  There is no CPU path.
"""
import ctypes as C
import math
import os
import threading

import numpy as np
import torch

from . import _lib

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8      # tf.train.AdamOptimizer defaults
BATCH_FIELDS = ("user_1hop", "user_2hop", "item_1hop", "item_2hop",
                "target_user", "target_item", "label", "length")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _usable_cpus():
    """host cores this process may really use: the affinity mask AND the cgroup CPU quota (a container with 16 cores of
    quota on a 256-thread host reports 256 by affinity)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return max(1, n)


def batch_shapes(cfg, B):
    T, K, Fu, Fi = cfg.max_time_len, cfg.obj_per_time_slice, cfg.user_fnum, cfg.item_fnum
    return ((B, T, K, Fi), (B, T, K, Fu), (B, T, K, Fu), (B, T, K, Fi), (B, Fu), (B, Fi), (B,), (B,))


def carve_batch(flat, shapes):
    """the eight int32 tensors of a batch as views of one flat buffer (every tensor 16-B aligned)"""
    out, off = [], 0
    for sh in shapes:
        n = int(np.prod(sh))
        out.append(flat[off:off + n].view(*sh))
        off += (n + 3) & ~3
    return out


def flat_batch_size(shapes):
    return sum((int(np.prod(sh)) + 3) & ~3 for sh in shapes)


class DeviceBatch(object):
    """int32 device tensors of one batch + the C struct pointing at them.  The eight tensors are views of ONE flat
    allocation (`flat`): a host batch goes up in a single copy, and a captured step refreshes its static batch with
    a single device-to-device copy."""

    def __init__(self, model, batch_data):
        if isinstance(batch_data, DeviceBatch):
            raise TypeError("already a DeviceBatch")
        if len(batch_data) != 8:
            raise ValueError("batch_data must be the 8-tuple of graph_loader.py:383")
        max_len = None
        on_device = all(torch.is_tensor(x) and x.device.type == "cuda" for x in batch_data)
        B = int(batch_data[6].shape[0]) if hasattr(batch_data[6], "shape") else len(batch_data[6])
        if B == 0:
            raise ValueError("empty batch")
        shapes = batch_shapes(model.cfg, B)
        self.B = B
        n_flat = flat_batch_size(shapes)

        def bad(i, got):
            return ValueError("batch_data[%d] (%s) has shape %s, expected %s" % (i, BATCH_FIELDS[i], tuple(got), shapes[i]))
        if on_device:
            self.flat = torch.empty((n_flat,), dtype=torch.int32, device=model.device)
            self.tensors = carve_batch(self.flat, shapes)
            for i, (dst, src) in enumerate(zip(self.tensors, batch_data)):
                if tuple(src.shape) != shapes[i]:
                    raise bad(i, src.shape)
                dst.copy_(src)                              # (dtype / device conversion included)
            max_len = int(self.tensors[7].max().item())     # one read-back per batch object
        else:
            # the whole feed tuple into ONE pinned int32 staging buffer, then one asynchronous copy to the device.  Nested
            # lists (what the reference's loader yields: ints, with float 0.0 in dummy slices, graph_loader.py:90-91) are
            # walked in C (_listpack: ~15x faster than np.asarray on lists, and on several native threads without the
            # GIL -- the walk is one cache miss per boxed int); arrays / host tensors are copied
            pinned, slot = model._staging(B, n_flat)
            views = carve_batch(pinned, shapes)
            lp = _lib.listpack()
            nthreads = int(getattr(model, "feed_threads", 1))
            try:
                listed = [i for i, x in enumerate(batch_data) if lp is not None and isinstance(x, (list, tuple))]
                if listed:      # every list-shaped tensor of the tuple in ONE threaded region (threads are created once)
                    try:
                        lp.pack_many([(batch_data[i], views[i].numpy(), shapes[i]) for i in listed], nthreads)
                    except ValueError as e:
                        i = listed[getattr(e, "tensor_index", 0)]
                        raise bad(i, np.asarray(batch_data[i]).shape)
                for i, (dst, x) in enumerate(zip(views, batch_data)):
                    if i in listed:
                        continue
                    a = x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
                    if tuple(a.shape) != shapes[i]:
                        raise bad(i, a.shape)
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(a.astype(np.int32, copy=False))))
                max_len = int(views[7].max()) if B else 0
                self.flat = torch.empty((n_flat,), dtype=torch.int32, device=model.device)
                self.flat.copy_(pinned, non_blocking=True)
            finally:
                # the slot may be rewritten once this copy has run (or at once, if the conversion raised)
                slot[1] = torch.cuda.current_stream(model.device).record_event()
                slot[2] = False
            self.tensors = carve_batch(self.flat, shapes)
        self.active_slices = active_slices(model, max_len)
        self.struct = _lib.Batch(*[_ptr(t) for t in self.tensors], B, self.active_slices)

    @classmethod
    def empty(cls, model, B, active=0):
        """uninitialised batch of B samples (a loader / a captured step fills it)"""
        self = cls.__new__(cls)
        shapes = batch_shapes(model.cfg, B)
        self.flat = torch.empty((flat_batch_size(shapes),), dtype=torch.int32, device=model.device)
        self.tensors = carve_batch(self.flat, shapes)
        self.B = B
        self.active_slices = int(active)
        self.struct = _lib.Batch(*[_ptr(t) for t in self.tensors], B, self.active_slices)
        return self


def active_slices(model, max_len):
    """score_batch_t.active_slices for a batch whose longest sample has `max_len` slices: the slices past every
    sample's length are masked out of the result by the model itself (dynamic_rnn's sequence_length,
    score.py:205-208; the attention mask, :182-185) and are not gathered or computed.  0 = all T."""
    T = int(model.cfg.max_time_len)
    if not getattr(model, "skip_masked_slices", True):
        return 0
    a = min(max(int(max_len), 1), T)
    return 0 if a >= T else a


class SCOREBASE(object):
    model_type = None

    def __init__(self, feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice,
                 user_fnum, item_fnum, seed=1111, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("score_amd needs an AMD GPU (HIP); there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.cfg = _lib.make_config(feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice,
                                    user_fnum, item_fnum, self.model_type)
        self.obj_per_time_slice = obj_per_time_slice
        self.entries, self.n_w, self.n_reg = _lib.param_layout(self.cfg)
        N, D = int(feature_size), int(eb_dim)
        f32 = dict(dtype=torch.float32, device=self.device)
        # emb_mtx, its two Adam slots and its gradient: ONE allocation, four [rows, D] views.  The dense optimizer sweep
        # streams all four together; carved from one block it measured 5-8 % faster than from four separate
        # allocations (tools/adam_layout_probe.py: 0.44-0.47 vs 0.47-0.51 ms at cfg-3)
        rows = self._table_rows(N)
        self._table_block = torch.empty((4, rows, D), **f32)
        self._tbl = self._table_block[0]
        # time-tiled table optimizer (include/score_hip.h, score_adam_table_t): live rows without a gradient are
        # updated the next time they are needed (or once per `adam_window` steps) instead of every step.  Bit-identical
        # to the per-step sweep wherever the table is observed; 0 = sweep the whole table every step.  24: at cfg-3 the
        # slice of a step (1/24 of 1.53 M rows) is through before the scatter starts; 12 - 22 measure 0.03 - 0.06 ms/step
        # slower, 26 - 32 the same, 36+ slower again (profiles/r02_probes.md)
        self._adam_window = int(os.environ.get("SCORE_ADAM_WINDOW", "24"))
        # ... and only where the sweep is worth replacing: its six streams over the table against two extra scans of
        # the state bytes and three more launches per step (cfg-2's 62 MB table: 0.391 ms/step swept, 0.425 tiled;
        # the reference's own shape, 587 MB: 0.513 -> 0.466; cfg-3, 2.35 GB: 1.76 -> 1.48)
        self.adam_tiled_min_bytes = int(os.environ.get("SCORE_ADAM_TILED_MIN_BYTES", str(256 << 20)))
        self._adam_dirty = False     # live rows may lag behind self.step (row_step says by how much)
        self._tiled_ready = False    # row_step / alpha_ring describe the table
        self._tiled = None           # (row_step, alpha_ring, score_adam_table_t)
        self._pending_sweep = None   # (row range, step, event) of the window slice not launched yet
        self._ev_stage = None        # a stage boundary of the backward pass (where the window slice starts)
        self.catchup_events = None   # optional (start, end) torch events around score_adam_catchup_ids (bench.py)
        self._ev_sweep = None
        self._ev_b4 = None           # stage boundary 4 of the backward pass (row scatter done): where the look-ahead catch-up starts
        self._ahead = None           # (DeviceBatch, event): its rows were brought up to date through the step in flight
        self._grads_pending = None   # event behind the dense gradient's finishers on the side stream (score_state_t.grads_done_event)
        self._ev_grads = None
        self._ev_loss = None
        # from this many (b, t) rows on, score_backward's end-of-pass finishers run on the side stream under the touched-row
        # update (cfg-3: ~40 us off the launch stream); below, the step is bound by the host's launch calls and the table's and
        # the dense variables' updates stay ONE launch behind finishers on the launch stream
        self.overlap_finishers_min_rows = 6144       # (the CCMR shape's 7,600 rows gain 1.7 % from it, the Tmall default's 2,200 lose 4 %)
        self.dense_adam_on_side = True       # (with the finishers overlap) the dense variables' ApplyAdam on the host's side stream
        self.loss_on_side = True            # the loss reduction of a training step on the engine's side stream (score_state_t.loss_done_event)
        self._early_loss = None      # set for the length of a train() call: the loss copied out right behind the forward pass
        self._early_loss_state = {"stream": None, "host": None, "event": None}
        self._sweep_st = None        # stream of the window slice when it runs beside the forward pass (adam_sweep_at = "f1")
        self._fwd_stage_event = None
        self._pinned_stream = self._pinned_handle = None
        self._row_list = None
        self._dense_pending = None      # event behind the dense variables' ApplyAdam when it ran on the side stream (apply_adam)
        self._ps_form = {}
        self._train_stream = None        # the stream the last forward_backward ran on (who else must wait for side-stream updates)
        self._plan_ready = None          # (batch, event, workspace, active slices) of an index plan launched a step ahead
        self._st_ahead = None
        self.plan_ahead = True           # apply_adam(next_batch=) also sorts the next batch's occurrences (side stream, behind the scatter)
        self._ev_arrays = {}
        self._alpha_memo = (None, 0.0)
        self._st_cache = {}
        self._evs = {}
        self._ps_last = False           # the last forward_backward ran as the per-sample whole-model kernels
        self._inline_on = False         # debug_flags bit 12 was set at the last forward_backward: every stream below IS the launch stream
        self._sweep_on_side = False     # the pending window slice was queued on self._side (by the one-call step)
        # the one-call step may alternate two PLAN buffers so that the next batch's sort needs nothing of this step and runs beside
        # its passes instead of behind its scatter.  Whether that pays depends on which cycle bounds the step -- pull -> look-ahead ->
        # sort -> pull or the launch stream's own chain, which the early sort then slows: "auto" decides by the number of
        # occurrences per batch (_two_buffers, TWO_BUFFERS_BELOW: a rule, the same on every run and rank); True / False force one.
        self.plan_two_workspaces = "auto"
        self._plan_events = [None, None]
        self._plan_stream = None
        self._fin_early = False
        self._ev_loss_dev = None
        self.fast_step = True           # train / train_async: the steady-state step of the per-sample form as one library call (_train_step_fast)
        self._step_args = self._step_T = self._step_side = None
        self._pb_cache = {}
        self._ev_dense = None
        self.w = torch.zeros((self.n_w,), **f32)
        self._alloc_optimizer()
        self._ws = {}              # (B, slot) -> (layout, buffer), least recently used first
        self.max_workspaces = 8
        ctx = C.c_void_p(0)
        _lib.check(self.lib.score_context_create(C.byref(ctx)), "score_context_create")
        self._ctx = ctx            # side stream + events of this model's score_forward/backward calls
        self.step = 0
        self.beta1_power = np.float32(ADAM_B1)
        self.beta2_power = np.float32(ADAM_B2)
        self.row0 = np.zeros((D,), dtype=np.float32)   # value of the masked variable row (score.py:44-47)
        self._drop_seed = int(seed)
        self.fwd_events = self.bwd_events = None
        self.scatter_mode = 0      # 0: sorted pull-form scatter, 1: float atomics (score_hip.h)
        self.global_batch = 0      # >0: the loss mean runs over this many samples (data parallel)
        self._side = None
        self.skip_masked_slices = True   # batches carry active_slices = max(length): slices every sample masks are skipped
        # per-step scalars in device memory (score_step_scalars_t): what a captured step reads its alpha / dropout seed from
        # sticky device word the kernels OR a bit into when a fed id lies outside the table (score_state_t.id_status)
        # word 0: the bits; word 1: optimizer steps the device suppressed because of it (score_guard_t.skipped)
        self._id_status = torch.zeros((2,), dtype=torch.int32, device=self.device)
        if self._guard_on:
            a = self._id_status.data_ptr()
            self._guard_dense = _lib.Guard(id_status=a, skipped=a + 4)      # ONE counted call per step: the dense variables'
            self._guard_table = _lib.Guard(id_status=a, skipped=None)
        else:
            self._guard_dense = self._guard_table = None
        self._scalars = torch.zeros((4,), dtype=torch.int32, device=self.device)
        # pinned staging ring for them: with a captured step the host runs many steps ahead of the GPU, so the slot a
        # queued H2D copy reads from must not be rewritten before that copy has run (one event per slot says when)
        self._scalars_ring = None
        self._scalars_slot = 0
        self._use_dev_scalars = False
        self._graph_on, self._graphs = False, {}
        self.debug_flags = 0       # score_state_t.debug_flags (A/B switches; bit 0: step-by-step H = 256 recurrence, bit 1: head forward in one launch, bit 2: f32-MFMA H = 128 recurrence, bit 3: tiled instead of panel GEMMs for the GRU projections, bit 4: panel form for their input gradients too, bit 6: the head layer by layer, bit 7: the temporal attention layer by layer)
        self.gemm_mode = 1         # 1: bf16x3 split (fp32-accurate) on the shapes where it measured faster, 0: f32 MFMA only
        # host feed path (nested lists / arrays -> pinned staging -> device): native threads of the list walk, and a ring
        # of pinned staging buffers per batch size (a buffer is reused once the H2D copy that read it has run)
        # (threads: the walk is one cache miss per boxed int, so it scales with cores until the memory system is busy --
        #  cfg-3 batch on the GPU box's host: 8.9 ms with 1 thread, 2.5 with 4, 1.6 with 8, 1.35 with 16)
        # where the time-tiled optimizer's window slice starts: "2" (stage boundary 2 of the backward pass, beside the
        # recurrence), "1" / "3" / "4", "plan" (behind the occurrence sort), "f1" (behind the fused gather, on a stream of its
        # own); "auto": "2" for the per-sample form (what its one-call step queues) and "plan" for the layer-by-layer pass --
        # since round 6 moved the recurrences' weight-gradient products beside the scatter, the backward pass has no slack
        # left for a 100-us HBM-bound scan beside its input-gradient product (cfg-3, one box, interleaved: plan 892 - 895 k
        # samples/s, "2" 884 - 888 k, "1" 874 - 886 k, "3" 858 - 864 k, "4" 854 - 861 k, "f1" 850 k: profiles/r06_probes.md);
        # the touched-row update from the plan's unique-row list instead of a state-byte scan
        # (measured +8 us net: off).  Both change WHERE work runs, never a result (tests/test_gpu_adam_tiled.py)
        self.adam_sweep_at = "auto"
        self.adam_touched_list = False
        self.feed_threads = int(os.environ.get("SCORE_FEED_THREADS", str(max(1, min(16, _usable_cpus())))))
        self._stage, self._stage_lock = {}, threading.Lock()
        self._init_params(seed)

    STAGING_SLOTS = 4
    # the optimizer kernels read score_state_t.id_status and apply nothing while it is set (score_guard_t: TF raises inside
    # sess.run for an id outside the table and no variable is updated, score.py:51-66,101-116).  Off for a row shard: the
    # sharded path rejects such a batch on the host, on every rank, before the step starts (score_amd/dist.py)
    _guard_on = True

    def _staging(self, B, n_flat):
        """-> (pinned int32 [n_flat] whose padding words are zero, its ring slot [tensor, copy-done event, busy])"""
        with self._stage_lock:
            ring = self._stage.get(B)
            if ring is None or ring[0][0][0].numel() != n_flat:
                # (zeroed once: the 16-byte alignment padding between the eight tensors is never written afterwards)
                ring = self._stage[B] = [[[torch.zeros((n_flat,), dtype=torch.int32).pin_memory(), None, False]
                                         for _ in range(self.STAGING_SLOTS)], 0]
                while len(self._stage) > 4:
                    self._stage.pop(next(iter(self._stage)))
            for _ in range(self.STAGING_SLOTS):
                slot = ring[0][ring[1]]
                ring[1] = (ring[1] + 1) % self.STAGING_SLOTS
                if not slot[2]:
                    break
            else:                                   # every slot is being filled by another thread: a private buffer
                slot = [torch.zeros((n_flat,), dtype=torch.int32).pin_memory(), None, False]
            slot[2] = True
        if slot[1] is not None:
            slot[1].synchronize()
        return slot[0], slot

    def feed(self, batches, depth=2):
        """Iterate over `batches` (feed tuples as the reference's GraphLoader yields them, graph_loader.py:383,397)
        `depth` batches AHEAD: a worker thread turns the next tuples into DeviceBatch objects -- list walk on native
        threads without the GIL, pinned staging, H2D copy on a stream of its own -- while the caller trains on the
        current one, so `for b in model.feed(loader): model.train(sess, b, lr, reg)` hides the host ingestion behind
        the step.  Yields DeviceBatch objects (train / eval take them as they take the tuples); results are those of
        feeding the tuples directly."""
        import queue
        q = queue.Queue(maxsize=max(1, int(depth)))
        stop = threading.Event()
        END = object()

        def hand_over(item):
            """False once the consumer has left (it drains the queue then: nothing may block on it for ever)"""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                torch.cuda.set_device(self.device)
                st = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(st):
                    for b in batches:
                        db = self.device_batch(b)
                        if not hand_over((db, st.record_event())):
                            return
                hand_over((END, None))
            except BaseException as e:              # handed to the consumer, raised there
                hand_over((e, None))
        t = threading.Thread(target=work, name="score-feed", daemon=True)
        t.start()
        try:
            while True:
                db, ev = q.get()
                if db is END:
                    return
                if isinstance(db, BaseException):
                    raise db
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                db.flat.record_stream(cur)          # (allocated on the worker's stream, consumed on this one)
                yield db
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)

    def _table_rows(self, feature_size):
        """rows of emb_mtx this object holds (a row shard overrides it, score_amd/dist.py)"""
        return feature_size

    # emb_mtx and its Adam slots as everybody outside the training step sees them: brought up to date first
    @property
    def table(self):
        self._flush_adam()
        return self._tbl

    @table.setter
    def table(self, t):
        self._flush_adam()
        self._tbl, self._tiled, self._tiled_ready = t, None, False

    # the flat dense variables and their Adam slots.  Their ApplyAdam may have run on the side stream (apply_adam: behind the dense
    # gradient's finishers, beside the table's touched-row update): whoever reads or writes them through here waits for it first
    @property
    def w(self):
        self._join_dense()
        return self._w

    @w.setter
    def w(self, t):
        self._w = t

    @property
    def w_m(self):
        self._join_dense()
        return self._w_m

    @w_m.setter
    def w_m(self, t):
        self._w_m = t

    @property
    def w_v(self):
        self._join_dense()
        return self._w_v

    @w_v.setter
    def w_v(self, t):
        self._w_v = t

    def _join_dense(self):
        ev, self._dense_pending = self._dense_pending, None
        if ev is not None:
            self._grads_pending = None          # (the update ran behind the finishers: they are through as well)
            cur = self._cur()
            cur.wait_event(ev)
            # (touched from inside a `with torch.cuda.stream(x)` block of the caller: x waits above, and so must the stream the
            #  training sequence runs on -- its next forward pass reads the variables too, and the pending event is gone by then)
            own = self._train_stream
            if own is not None and own.cuda_stream != cur.cuda_stream:
                own.wait_event(ev)

    @property
    def w_g(self):
        """the flat dense gradient of the last backward pass.  Its finishers (slab reduce, column sums) may still be running
        on the engine's side stream (score_state_t.grads_done_event): whoever reads it through here waits for them first"""
        self._join_grads()
        return self._w_g

    def _join_grads(self):
        ev, self._grads_pending = self._grads_pending, None
        if ev is not None:
            cur = self._cur()
            cur.wait_event(ev)
            own = self._train_stream
            if own is not None and own.cuda_stream != cur.cuda_stream:
                own.wait_event(ev)

    @property
    def adam_window(self):
        return self._adam_window

    @adam_window.setter
    def adam_window(self, k):
        # a new window restarts the slice schedule: nothing may be owed across the change (a row could otherwise wait
        # old + new window steps for its slice, past what the alpha ring remembers)
        if int(k) != self._adam_window:
            self._flush_adam()
            self._adam_window = int(k)

    @property
    def table_m(self):
        self._flush_adam()
        return self._tbl_m

    @property
    def table_v(self):
        self._flush_adam()
        return self._tbl_v

    def __del__(self):
        # Work this object queued on its OWN streams (the dense variables' ApplyAdam and the index plan on the host's side stream,
        # the window slice, the early loss copy) may still be running: the caching allocator knows a tensor only by the stream
        # it was allocated on and would hand the dying model's buffers to the next allocation while those kernels still write
        # them (seen as two "identical" models diverging by a few ulp when one of them inherited such a block).  Wait first.
        for st in (getattr(self, "_side", None), getattr(self, "_sweep_st", None), getattr(self, "_plan_stream", None),
                   (getattr(self, "_early_loss_state", None) or {}).get("stream")):
            try:
                if st is not None:
                    st.synchronize()
            except Exception:
                pass
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx is not None and ctx.value and getattr(self, "lib", None) is not None:
            try:
                self.lib.score_context_destroy(ctx)      # (synchronises the engine's side stream)
            except Exception:
                pass

    # ------------------------------------------------------------------ parameters
    def _alloc_optimizer(self):
        f32 = dict(dtype=torch.float32, device=self.device)
        self._tbl_m, self._tbl_v, self.table_g = self._table_block[1], self._table_block[2], self._table_block[3]
        self._table_block[1:].zero_()
        # per-row optimizer state byte (score_adam_rows): 0 = moments zero, 1 = live, 2 = gradient this step
        self.table_flags = torch.zeros((self._tbl.shape[0],), dtype=torch.uint8, device=self.device)
        self._row_grads = False      # table_g holds valid rows only where table_flags == 2
        self._flags_marked = False   # table_flags may hold 2s
        self.w_m = torch.zeros((self.n_w,), **f32)
        self.w_v = torch.zeros((self.n_w,), **f32)
        # four spare floats behind the dense gradient: the sharded path appends its share of the log-loss so that
        # one all-reduce carries both (score_amd/dist.py)
        self._w_g_ext = torch.zeros((self.n_w + 4,), **f32)
        self._w_g = self._w_g_ext[:self.n_w]

    def _view(self, flat, entry):
        name, off, rows, cols, _, _ = entry
        n = rows * (cols if cols else 1)
        v = flat[off:off + n]
        return v.view(rows, cols) if cols else v

    def _init_table(self, seed, row_stride=1, row_first=0, n_global=None):
        """score_table_init: element (global row, col) is a pure function of the seed, so every sharding of a seed
        holds the same table and a shard initialises only its own rows.  Row 0 is the masked dummy row."""
        _lib.check(self.lib.score_table_init(_ptr(self.table), self.table.shape[0], self.table.shape[1], int(row_stride),
                                             int(row_first), int(n_global if n_global is not None else self.table.shape[0]),
                                             C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), self._stream()),
                   "score_table_init")
        self.row0 = np.zeros((self.table.shape[1],), dtype=np.float32)

    def _init_params(self, seed):
        """TF initialiser families: truncated normal(0,1) table (score.py:44), glorot-uniform
        dense/GRU kernels, GRU gate bias 1, zeros elsewhere."""
        self._init_table(seed)
        self._init_dense(seed)

    def _init_dense(self, seed):
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(seed))
        for e in self.entries:
            v = self._view(self.w, e)
            if e[5] == 2:
                lim = math.sqrt(6.0 / (e[2] + e[3]))
                v.uniform_(-lim, lim, generator=gen)
            elif e[5] == 1:
                v.fill_(1.0)
            else:
                v.zero_()

    def _table_host(self):
        """the rows of emb_mtx this object holds, as the variable stores them (row 0 with its masked value)"""
        t = self.table.cpu().numpy()
        t[0] = self.row0
        return t

    def _table_load(self, emb):
        emb = np.asarray(emb, dtype=np.float32)
        if emb.shape != tuple(self.table.shape):
            raise ValueError("emb_mtx shape %s != %s" % (emb.shape, tuple(self.table.shape)))
        self.row0 = emb[0].copy()
        self.table.copy_(torch.from_numpy(emb))
        self.table[0].zero_()

    def _set_dense(self, params):
        for e in self.entries:
            v = self._view(self.w, e)
            a = np.asarray(params[e[0]], dtype=np.float32).reshape(tuple(v.shape))
            v.copy_(torch.from_numpy(a))

    def get_params(self):
        """dict TF variable name -> ndarray (emb_mtx carries its masked row 0 value)."""
        out = {"emb_mtx": self._table_host()}
        for e in self.entries:
            out[e[0]] = self._view(self.w, e).cpu().numpy().copy()
        return out

    def set_params(self, params):
        self._table_load(params["emb_mtx"])
        self._set_dense(params)

    def dense_table_grad(self):
        """[N, D] gradient of the last forward_backward (device tensor)."""
        if self._row_grads:
            return self.table_g * (self.table_flags == 2).unsqueeze(1)
        return self.table_g

    def _drop_row_marks(self):
        # rows a previous backward marked (state 2) that no optimizer step consumed: back to "live"
        if self._flags_marked:
            if self._tiled is not None and self._tiled_ready:
                # (with row_step = the step they were brought up to before that pass: a row that had been in state 0 would
                #  otherwise keep a stale count and look more steps behind than the alpha ring remembers)
                _lib.check(self.lib.score_adam_unmark(C.byref(self._tiled[2]), int(self.step), self._stream()),
                           "score_adam_unmark")
            self.table_flags.clamp_(max=1)
        self._flags_marked = False
        self._row_grads = False
        self._row_list = None        # (the unique-row list of a plan describes THAT backward pass's marks only)

    def _begin_row_grads(self):
        self._drop_row_marks()
        self._row_grads = True
        self._flags_marked = True

    def refresh_row_flags(self):
        """Recompute the row state bytes after the Adam slots were set from outside."""
        live = (self.table_m != 0).any(dim=1) | (self.table_v != 0).any(dim=1)
        self.table_flags.copy_(live.to(torch.uint8))
        self._row_grads = False
        self._flags_marked = False
        self._tiled_ready = False

    def get_dense_grads(self):
        """Gradients of the dense variables only (no [N, D] host copy of the table's)."""
        return {e[0]: self._view(self.w_g, e).cpu().numpy().copy() for e in self.entries}

    def get_grads(self):
        out = {"emb_mtx": self.dense_table_grad().cpu().numpy()}
        out.update(self.get_dense_grads())
        return out

    # ------------------------------------------------------------------ device plumbing
    def _workspace(self, B, slot=0):
        """One workspace per (batch size, slot); slots let the sharded path plan batch t+1 while
        batch t is still computing."""
        key = (B, slot)
        ent = self._ws.pop(key, None)
        if ent is None:
            self._pb_cache = {}
            lay = _lib.workspace_layout(self.cfg, B)
            buf = torch.empty((lay.total_bytes // 4,), dtype=torch.float32, device=self.device)
            # least-recently-used eviction, one entry at a time.  Whoever still needs an evicted buffer (an index
            # plan in flight, score_amd/dist.py) holds its own reference to it, so dropping the cache entry never
            # pulls memory from under a step; the device-wide wait keeps the allocator from handing the block to a
            # new tensor while kernels enqueued on other streams still use it (evictions are rare: a new batch size).
            while len(self._ws) >= self.max_workspaces:
                torch.cuda.synchronize(self.device)
                self._ws.pop(next(iter(self._ws)))
            ent = (lay, buf)
        self._ws[key] = ent          # most recently used last
        return ent

    def _state(self, ws, joined=True):
        # (the property: joins a dense ApplyAdam still running on a side stream; joined = False: the caller orders the streams itself)
        w = self.w if joined else self._w
        key = (ws.data_ptr(), self._tbl.data_ptr(), w.data_ptr(), self.table_flags.data_ptr())
        ent = self._st_cache.get(key)
        if ent is None:
            if len(self._st_cache) > 16:
                self._st_cache.clear()
            ent = self._st_cache[key] = _lib.State(
                table=_ptr(self._tbl), n_table_rows=self._tbl.shape[0], w=_ptr(w), workspace=_ptr(ws),
                workspace_bytes=ws.numel() * 4, context=self._ctx, id_status=_ptr(self._id_status))
        # one struct per (workspace, table, variables), refreshed in place: building it anew was ~8 us of Python per step
        st = ent
        st.workspace_bytes = ws.numel() * 4          # (a new workspace may sit where an evicted one of another size did)
        st.n_table_rows = self._tbl.shape[0]
        st.scatter_mode = int(self.scatter_mode)
        st.global_batch = int(self.global_batch)
        st.gemm_mode = int(self.gemm_mode)
        st.debug_flags = int(self.debug_flags)
        st.row_flags = self.table_flags.data_ptr() if self.scatter_mode == 0 else None
        st.step_scalars = self._scalars.data_ptr() if self._use_dev_scalars else None
        st.id_status = self._id_status.data_ptr()      # (a caller may have pointed the struct at a status word of its own: dist.py)
        st.gather_done_event = st.plan_done_event = st.grads_done_event = st.loss_done_event = st.loss_host = None
        st.plan_workspace = None
        return st

    def _event_array(self, events):
        """torch.cuda.Events (timing enabled, already recorded once) -> hipEvent_t[], or NULL."""
        if not events:
            return None
        key = tuple(id(e) for e in events)
        ent = self._ev_arrays.get(key)
        if ent is None or any(a is not b for a, b in zip(ent[1], events)):
            if len(self._ev_arrays) > 16:
                self._ev_arrays.clear()
            arr = (C.c_void_p * len(events))(*[C.c_void_p(e.cuda_event if e is not None else 0) for e in events])
            ent = self._ev_arrays[key] = (arr, list(events))      # (the list keeps the events alive: ids are not reused meanwhile)
        return ent[0]

    def enable_stage_events(self, on=True):
        """Record stage-boundary events inside score_forward/backward (bench.py's live
        per-kernel timing).  See include/score_hip.h for what each boundary brackets."""
        if not on:
            self.fwd_events = self.bwd_events = None
            return
        mk = lambda n: [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        self.fwd_events, self.bwd_events = mk(5), mk(6)
        for e in self.fwd_events + self.bwd_events:
            e.record()          # forces creation of the underlying hipEvent_t

    def _rec(self, name, stream):
        """record the model's reusable event `name` on `stream` and return it.  One event object per purpose, re-recorded every step
        (a new torch Event per record is a hipEventCreate / hipEventDestroy pair: five of them per step were ~25 us of a host-bound
        200-us step).  Safe because every wait on such an event is ISSUED before its next record, and a stream wait binds to the
        record that is current when it is issued."""
        ev = self._evs.get(name)
        if ev is None:
            # (stream-to-stream ordering only: an event without the system-scope fence of torch.cuda.Event -- _lib.DevEvent; the
            #  one event the HOST reads results behind, the early loss copy's, stays an ordinary one)
            ev = self._evs[name] = self._ev_new(host=name in self._HOST_READ_EVENTS)
        ev.record(stream)
        return ev

    _HOST_READ_EVENTS = ("early_loss",)
    device_events = True            # False: torch.cuda.Event everywhere (the A/B of profiles/r06_probes.md section 9)

    def _ev_new(self, host=False):
        """an event for stream-to-stream ordering (_lib.DevEvent: no system-scope fence at its records), or -- host=True: the host
        reads results behind it -- an ordinary torch.cuda.Event"""
        return torch.cuda.Event() if (host or not self.device_events) else _lib.DevEvent()

    def _cur(self):
        """torch.cuda.current_stream(self.device), looked up once per public call: the query costs ~4 us of Python and a
        training step asked for it eleven times (the small shapes are host-bound).  `_pinned_stream` is set by the
        entry points below for the duration of one call; a `with torch.cuda.stream(...)` block inside clears it."""
        c = self._pinned_stream
        return c if c is not None else torch.cuda.current_stream(self.device)

    def _stream(self):
        c = self._pinned_stream
        if c is not None:
            return self._pinned_handle
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    class _Pin(object):
        """context: look the current stream up once and hold it for the calls inside"""

        def __init__(self, m):
            self.m = m

        def __enter__(self):
            m = self.m
            self.prev = (m._pinned_stream, m._pinned_handle)
            if m._pinned_stream is None:
                cur = torch.cuda.current_stream(m.device)
                m._pinned_stream, m._pinned_handle = cur, C.c_void_p(cur.cuda_stream)
            return m

        def __exit__(self, *a):
            self.m._pinned_stream, self.m._pinned_handle = self.prev

    class _Unpin(object):
        """context: inside a `with torch.cuda.stream(other)` block the current stream is the other one"""

        def __init__(self, m):
            self.m = m

        def __enter__(self):
            m = self.m
            self.prev = (m._pinned_stream, m._pinned_handle)
            m._pinned_stream = m._pinned_handle = None

        def __exit__(self, *a):
            self.m._pinned_stream, self.m._pinned_handle = self.prev

    def device_batch(self, batch_data):
        return batch_data if isinstance(batch_data, DeviceBatch) else DeviceBatch(self, batch_data)

    def persample_form(self, B, active_slices=0):
        """True if a batch of B samples with `active_slices` computed slices runs as the per-sample whole-model kernels
        (include/score_hip.h score_persample_form; csrc/persample.h): the reference's own shapes.  The step is then a handful of
        launches whose COUNT on the launch stream is what it costs, and everything off the scatter's chain goes to side streams."""
        key = (int(B), int(active_slices), int(self.debug_flags), int(self.scatter_mode), int(self.global_batch))
        got = self._ps_form.get(key)
        if got is None:
            _, ws = self._workspace(B)
            st = self._state(ws)
            got = self._ps_form[key] = self.lib.score_persample_form(C.byref(self.cfg), C.byref(st), int(B), int(active_slices)) == 1
        return got

    def gemm_forms(self, B, active_slices=0):
        """(x_form, dx_form) of include/score_hip.h score_gemm_forms: how the GRU input projections / their input gradients
        of a batch of B samples run under this model's gemm_mode and debug_flags (0 = tiled kernels, n = panel groups per side)"""
        _, ws = self._workspace(B)
        st = self._state(ws)
        xf, df = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.score_gemm_forms(C.byref(self.cfg), C.byref(st), int(B), int(active_slices), C.byref(xf),
                                             C.byref(df)), "score_gemm_forms")
        return xf.value, df.value

    def ws_tensor(self, B, field, shape):
        """View of a named workspace region (tests / introspection).  Per-slice regions hold
        [B * A, .] rows when the batch ran with A = active_slices < T: pass A in place of T."""
        lay, buf = self._workspace(B)
        off = getattr(lay, field)
        n = int(np.prod(shape))
        return buf[off:off + n].view(*shape)

    # ------------------------------------------------------------------ forward / backward / update
    def _forward(self, db, reg_lambda, keep_prob, masks, gather_event=None, sweep=False, stage_event=None,
                 loss_event=None):
        lay, ws = self._workspace(db.B)
        st = self._state(ws)
        if loss_event is not None:
            st.loss_done_event = C.c_void_p(loss_event.cuda_event)
        st.loss_host = None
        if self._early_loss is not None and self._ps_last and loss_event is not None:
            # train() on the per-sample form: the forward kernel's last workgroup stores the loss into pinned host memory itself
            # (score_state_t.loss_host); the caller reads it behind loss_event -- no copy, no stream of its own
            el = self._early_loss
            if el["host"] is None:
                el["host"] = torch.zeros((4,), dtype=torch.float32).pin_memory()
            st.loss_host = el["host"].data_ptr()
        if self._tiled_on():
            self._catchup(db, sweep)
        else:
            self._flush_adam()
        if gather_event is not None:
            st.gather_done_event = C.c_void_p(gather_event.cuda_event)
        m0 = m1 = None
        if masks is not None:
            m0 = torch.as_tensor(np.asarray(masks[0]), dtype=torch.uint8).to(self.device).contiguous()
            m1 = torch.as_tensor(np.asarray(masks[1]), dtype=torch.uint8).to(self.device).contiguous()
            if tuple(m0.shape) != (db.B, 200) or tuple(m1.shape) != (db.B, 80):
                raise ValueError("dropout masks must be [B,200] and [B,80]")
        seed = (self._drop_seed * 0x9E3779B1 + self.step * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
        events = self.fwd_events
        if stage_event is not None:          # (k, event): score_forward records it at its stage boundary k
            events = list(events) if events else [None] * 5
            if events[stage_event[0]] is None:
                events[stage_event[0]] = stage_event[1]
            self._fwd_stage_event = events[stage_event[0]]
        rc = self.lib.score_forward(C.byref(self.cfg), C.byref(st), C.byref(db.struct), float(reg_lambda),
                                    float(keep_prob), _ptr(m0), _ptr(m1), C.c_uint64(seed),
                                    self._event_array(events), self._stream())
        _lib.check(rc, "score_forward")
        self._keep = (m0, m1)
        return lay, ws, st

    def forward_backward(self, batch_data, reg_lambda, keep_prob=1.0, dropout_masks=None):
        """Forward + backward; gradients land in self.w_g (without the L2 term) and
        self.table_g (dense [N,D]).  Returns the device workspace layout/buffer."""
        with self._Pin(self):
            return self._forward_backward(batch_data, reg_lambda, keep_prob, dropout_masks)

    def _forward_backward(self, batch_data, reg_lambda, keep_prob, dropout_masks):
        db = self.device_batch(batch_data)
        plan_done = None
        cur = self._cur()
        self._train_stream = cur
        # debug_flags bit 12 (4096, score_hip.h): NO second stream anywhere -- the engine's forks, the index plan, the window
        # slice, the look-ahead, the dense ApplyAdam and the early loss copy all run on the launch stream, in launch order
        # (tests/test_gpu_model.py compares the overlap modes with it bit for bit; the first thing to try on a suspected race)
        inline = bool(int(self.debug_flags) & 4096)
        if inline != self._inline_on or (inline and self._side is not None and self._side.cuda_stream != cur.cuda_stream):
            torch.cuda.synchronize(self.device)
            self._side = self._sweep_st = self._early_loss_state["stream"] = None
            self._plan_ready = None
            self._inline_on = inline
        if self.scatter_mode == 0:
            if self._side is None:
                self._side = cur if inline else torch.cuda.Stream(device=self.device)
                self._side_handle = C.c_void_p(self._side.cuda_stream)
                self._ev_gather = self._ev_new()
                self._ev_gather.record(cur)          # materialise the hipEvent_t
        # the occurrence sort starts together with the forward (an event recorded here, not between the gather and
        # the input projections: a record between two launches costs ~5 us of bubble on the main stream, and the
        # latency-bound gather hardly notices the sort beside it: 1.489 -> 1.476 ms/step)
        early = self.scatter_mode == 0
        if early:
            ev_start = self._rec("start", cur)
        # time-tiled optimizer: this step's slice of the table (rows nobody in the batch touches: any time between the
        # batch rows' catch-up and the touched-row update will do).  "f1": behind the fused gather, i.e. beside the forward
        # recurrence, which is matrix-bound and fills half the CUs; 1 .. 4: at that stage boundary of the backward pass
        self._ps_last = self.persample_form(db.B, db.active_slices)
        sweep_at = str(self.adam_sweep_at)
        if sweep_at not in ("plan", "f1", "1", "2", "3", "4"):
            sweep_at = "2" if self._ps_last else "plan"      # ("auto"; an unknown value must not leave the window slice unlaunched)
        fwd_stage = None
        if sweep_at == "f1" and self._tiled_on():
            if self._ev_stage is None:
                self._ev_stage = self._ev_new()
                self._ev_stage.record(cur)              # materialise the hipEvent_t
            fwd_stage = (1, self._ev_stage)
        # the loss reduction (one workgroup, no reader inside the step) on the engine's side stream: score_backward's first
        # launch then follows the head directly (score_state_t.loss_done_event; score_backward joins that stream, so the
        # loss is final on this stream behind the pass)
        ev_loss = None
        # (only where the device, not the host's launch calls, bounds the step: tmall_default 0.330 -> 0.335 ms with it, cfg-3
        #  1.2811 -> 1.2762, four alternating pairs on one box)
        # (the per-sample kernels: always -- every launch taken off the chain is ~5 us there)
        side_ok = self._ps_last or db.B * (db.active_slices or int(self.cfg.max_time_len)) >= self.overlap_finishers_min_rows
        if self.loss_on_side and not self._graph_on and not self._use_dev_scalars and side_ok:
            # (train() reads the loss on the host behind this event -- from pinned memory the per-sample kernel wrote, or through a
            #  copy on another stream that only waits for it: an ordinary event then, with its system-scope release; train_async:
            #  it only orders streams, and a device event does)
            host_reads = self._early_loss is not None
            if host_reads:
                if self._ev_loss is None:
                    self._ev_loss = torch.cuda.Event()
                    self._ev_loss.record(cur)               # materialise the hipEvent_t
                ev_loss = self._ev_loss
            else:
                if self._ev_loss_dev is None:
                    self._ev_loss_dev = self._ev_new()
                    self._ev_loss_dev.record(cur)
                ev_loss = self._ev_loss_dev
        lay, ws, st = self._forward(db, reg_lambda, keep_prob, dropout_masks,
                                    gather_event=self._ev_gather if (self.scatter_mode == 0 and not early) else None,
                                    sweep=True, stage_event=fwd_stage, loss_event=ev_loss)
        if self._early_loss is not None and st.loss_host:
            self._early_loss["event"] = ev_loss
        elif self._early_loss is not None:
            # train(): the loss is final here, a whole backward pass and optimizer step before the stream is through -- it is
            # copied to pinned memory on a stream of its own behind this point, so the caller's read-back (score.py:101-116
            # returns the loss every step) waits for the forward only and the host goes on queueing
            el = self._early_loss
            if el["stream"] is None:
                el["stream"] = cur if inline else torch.cuda.Stream(device=self.device)
            if el["host"] is None:
                el["host"] = torch.zeros((4,), dtype=torch.float32).pin_memory()
            el["stream"].wait_event(ev_loss if ev_loss is not None else cur.record_event())
            with torch.cuda.stream(el["stream"]), self._Unpin(self):
                el["host"].copy_(ws[lay.loss:lay.loss + 4], non_blocking=True)
                el["event"] = self._rec("early_loss", el["stream"])
        if fwd_stage is not None and self._pending_sweep is not None:
            if self._sweep_st is None:
                self._sweep_st = cur if inline else torch.cuda.Stream(device=self.device)      # (its own stream: the occurrence sort must not queue behind it)
            self._sweep_st.wait_event(self._fwd_stage_event)
            self._launch_sweep(self._sweep_st)
        if self.scatter_mode == 0:
            # occurrence sort for the pull-form scatter: depends on the indices only.  It runs on a side
            # stream from the start of the forward (see above), under the gather / GRUs / attention / head and
            # the first half of the backward, and score_backward waits for it just before the row scatter
            # (its workspace regions are its own; the previous step's scatter, their last reader, is behind
            # the event the side stream waits for)
            want_list = self._tiled_on() and bool(self.adam_touched_list)
            pr, self._plan_ready = self._plan_ready, None
            plan_here = pr is not None and pr[0] is db and pr[3] == db.active_slices and not want_list and (
                pr[2] == ws.data_ptr() or pr[2] == self._plan_buffer(db.B, 1))
            if plan_here:
                if pr[2] != ws.data_ptr():
                    st.plan_workspace = pr[2]      # (sorted into the second plan buffer by the one-call step: score_state_t.plan_workspace)
                # apply_adam(next_batch=db) of the previous step has already sorted this batch's occurrences, behind that step's
                # row scatter (_plan_ahead): since the per-sample kernels, the six launches of the sort (~110 us with their gaps)
                # are longer than the forward and backward kernels they used to hide under
                plan_done = pr[1]
                row_list = None
                st.plan_done_event = C.c_void_p(plan_done.cuda_event)
                self._plan_done = plan_done
                early = None
        if self.scatter_mode == 0 and early is not None:
            self._side.wait_event(ev_start if early else self._ev_gather)
            # (dedup = 2: also the list of the batch's unique rows, for score_adam_touched_rows -- the touched-row update
            #  driven by that list instead of a scan of the table's state bytes.  OFF by default: measured on one box,
            #  alternating runs (tools/ab_env.sh), the update itself is 8 - 10 us shorter but the three extra plan kernels
            #  on the side stream cost the input projections / recurrence beside them 16 - 18 us: 1.306 vs 1.298 ms/step.
            #  model.adam_touched_list = True turns it on.)
            # (the side stream is handed to the call: entering and leaving a `with torch.cuda.stream(...)` block was ~10 us)
            _lib.check(self.lib.score_index_plan(C.byref(self.cfg), C.byref(st), C.byref(db.struct), 1,
                                                 2 if want_list else 0, self._side_handle), "score_index_plan")
            row_list = (lay, ws) if want_list else None
            plan_done = self._rec("plan", self._side)
            st.plan_done_event = C.c_void_p(plan_done.cuda_event)
            self._plan_done = plan_done                  # keep the event alive until the backward has run
        # time-tiled optimizer: where this step's slice of the table starts.  Beside the backward recurrence (stage
        # boundary 2 of score_backward: a 140-us VALU-bound kernel next to a matrix-bound one that fills half the CUs)
        # costs the step 0.02 ms; behind the occurrence sort -- beside the fused attention forward, whose 8-wave, 232-
        # register workgroups cannot share a CU with it -- 0.06; boundaries 1 / 3 / 4: 0.03 / 0.03 / 0.2
        # (model.adam_sweep_at = plan|1|2|3|4, profiles/r02_probes.md)
        if self._pending_sweep is None:
            sweep_at = ""
        if sweep_at == "plan":
            self._launch_sweep(self._side)
        if self.scatter_mode == 0:
            self._begin_row_grads()         # the pull kernels mark what they write; no zero fill
            self._row_list = row_list       # (the rows they will mark, listed by the plan)
        else:
            self._drop_row_marks()
            self.table_g.zero_()
        events = self.bwd_events
        ev_sweep_start = None
        if sweep_at in ("1", "2", "3", "4"):
            # the slice starts at a stage boundary of the backward pass (score_backward records the event there)
            if self._ev_stage is None:
                self._ev_stage = self._ev_new()
                self._ev_stage.record(cur)              # materialise the hipEvent_t
            events = list(events) if events else [None] * 6
            k = int(sweep_at)
            if events[k] is None:
                events[k] = self._ev_stage
            ev_sweep_start = events[k]
        self._b4_recorded = None
        self._b4_any = None
        if self._look_ahead and self.scatter_mode == 0 and (self._tiled_on() or self.plan_ahead):
            # boundary 4 (the row scatter has marked every row that gets this step's gradient): apply_adam(next_batch=)
            # starts the next batch's catch-up there, on the side stream
            if self._ev_b4 is None:
                self._ev_b4 = self._ev_new()
                self._ev_b4.record(cur)
            events = list(events) if events else [None] * 6
            if events[4] is None:
                events[4] = self._ev_b4
            self._b4_recorded = events[4]
            self._b4_any = events[4]
        self._join_grads()                    # (a previous pass's finishers: they wrote the buffer this pass writes)
        if (self.scatter_mode == 0 and not self._use_dev_scalars and not self._graph_on and side_ok
                and (self._tiled_on() or self.persample_form(db.B, db.active_slices))):
            if self._ev_grads is None:
                self._ev_grads = self._ev_new()
                self._ev_grads.record(cur)              # materialise the hipEvent_t
            st.grads_done_event = C.c_void_p(self._ev_grads.cuda_event)
            self._grads_pending = self._ev_grads
            # (score_backward then forks the finishers in FRONT of the row scatter, csrc/engine.hip "fin_early": the dense variables'
            #  update may not follow them on another stream any more -- the scatter reads the co-attention weights)
            self._fin_early = not self._ps_last and not (int(self.debug_flags) & 16384)
        else:
            self._fin_early = False
        rc = self.lib.score_backward(C.byref(self.cfg), C.byref(st), C.byref(db.struct), float(keep_prob),
                                     _ptr(self._w_g), _ptr(self.table_g), self._event_array(events),
                                     self._stream())
        _lib.check(rc, "score_backward")
        if ev_sweep_start is not None:
            self._launch_sweep(self._side, behind=ev_sweep_start)
        return lay, ws

    def _alpha(self, lr):
        key = (lr, self.step, float(self.beta1_power), float(self.beta2_power))
        if self._alpha_memo[0] != key:      # (asked for two or three times per step: ~2 us of NumPy scalars each)
            f = np.float32
            self._alpha_memo = (key, float(f(f(lr) * np.sqrt(f(1) - self.beta2_power) / (f(1) - self.beta1_power))))
        return self._alpha_memo[1]

    def apply_adam(self, lr, reg_lambda, next_batch=None):
        """tf.train.AdamOptimizer(lr).minimize(loss) update (score.py:96-99): dense over the
        whole table (the emb_mtx*mask gradient is dense) and over every dense variable.
        next_batch (a DeviceBatch, optional): the batch the NEXT step will train on, when the caller already holds it (the
        reference's loader keeps a queue of ten, graph_loader.py:280-281).  With the time-tiled table optimizer its rows
        are then brought up to date -- through this very step -- on the side stream beside this step's weight-gradient
        products, instead of in front of the next forward pass (score_adam_catchup_ids_through).  Same updates, same order
        per row: the same bits.  A different batch next is fine (it is caught up the usual way)."""
        with self._Pin(self):
            # (decided NOW: the first tiled step fills row_step on this stream inside _adam_table_tiled -- a look-ahead queued behind
            #  the scatter's event on the side stream would not be behind that fill)
            want_ahead = self._tiled_on() and self._row_grads and next_batch is not None and self._tiled_ready

            def side_work():
                if want_ahead:
                    self._catchup_ahead(next_batch, lr)
                # (the per-sample form only: at cfg-3 the library's sort of 2.9 M occurrences behind the scatter -- instead of beside the
                #  next forward pass's recurrences, which leave half the chip idle -- cost 1.19 -> 1.31 ms/step, two alternating pairs)
                if next_batch is not None and self.plan_ahead and self._ps_last:
                    self._plan_ahead(next_batch)
            # the per-sample form queues the optimizer FIRST: the step is bound by this thread's launch calls there, and the next
            # forward pass waits for the dense variables (through the weight images) longer than for the look-ahead
            if not self._ps_last:
                side_work()
            if self._tiled_on() and self._row_grads:
                if self._grads_pending is not None:
                    # the dense gradient's finishers are still running on the side stream: the table's touched rows (row
                    # gradients only) first, the dense variables behind the finishers' event -- on the host's side stream
                    # (idle by now: the look-ahead catch-up was started at boundary 4), so that the launch stream goes from the
                    # touched rows straight into the next step; whoever touches the dense variables next waits (self.w)
                    if self._ps_last:
                        # (the per-sample form: ONE launch for the touched rows and the dense variables, behind the finishers' event --
                        #  they end before the row scatter does, tools/kernel_sequence.sh)
                        self._join_grads()
                        if not self._adam_table_tiled(lr, dense=(reg_lambda,)):
                            self.adam_dense(lr, reg_lambda)
                        side_work()
                        self.adam_advance()
                        return
                    if self._fin_early:
                        # round 6: the finishers ran beside the scatter and are through by now -- the touched rows and the dense
                        # variables in ONE launch on the launch stream, behind the scatter (no stream hop in front of the next pass)
                        self._join_grads()
                        if not self._adam_table_tiled(lr, dense=(reg_lambda,)):
                            self.adam_dense(lr, reg_lambda)
                        self.adam_advance()
                        return
                    self._adam_table_tiled(lr)
                    # (the per-sample form: on the launch stream.  There the finishers are forked right behind the backward KERNEL, i.e.
                    #  in front of the row scatter -- and pull_kernel reads the co-attention weights: a dense update behind the
                    #  finishers' event alone, on another stream, races with it.  That was the ulp drift of two identical models under
                    #  tests/test_gpu_bad_ids.py earlier in round 5.  The layer-by-layer pass forks its finishers behind the scatter.)
                    side = self._side if (self.dense_adam_on_side and not self._ps_last) else None
                    if side is not None:
                        side.wait_event(self._grads_pending)
                        self.adam_dense(lr, reg_lambda, stream=side)
                        if self._ev_dense is None:
                            self._ev_dense = self._ev_new()
                        self._ev_dense.record(side)
                        self._dense_pending = self._ev_dense
                    else:
                        self._join_grads()
                        self.adam_dense(lr, reg_lambda)
                # (else the touched rows and the dense variables in one launch: nothing stands between them)
                elif not self._adam_table_tiled(lr, dense=(reg_lambda,)):
                    self.adam_dense(lr, reg_lambda)
            elif (self._row_grads and not self._use_dev_scalars and not self._adam_dirty and self._ps_last):
                # the per-step sweep of a small table (cfg-2) and the dense variables in ONE launch (score_adam_rows_and_dense)
                self._tiled_ready = False
                self._join_grads()
                _lib.check(self.lib.score_adam_rows_and_dense(
                    _ptr(self._tbl), _ptr(self._tbl_m), _ptr(self._tbl_v), _ptr(self.table_g), self._tbl.shape[0], self._tbl.shape[1],
                    _ptr(self.table_flags), _ptr(self.w), _ptr(self.w_m), _ptr(self.w_v), _ptr(self._w_g), self.n_w, self.n_reg,
                    float(reg_lambda), self._alpha(lr), ADAM_B1, ADAM_B2, ADAM_EPS, self._guard_dense, self._stream()),
                    "score_adam_rows_and_dense")
                self._row_grads = False
                self._flags_marked = False
            else:
                self.adam_table(lr)
                self.adam_dense(lr, reg_lambda)        # (reads self.w_g: waits for the finishers' event first)
            if self._ps_last:
                side_work()
            self.adam_advance()

    # ------------------------------------------------------------------ time-tiled table optimizer
    _tiled_supported = True          # (a subclass may opt out)

    def _tiled_on(self):
        return (self.adam_window > 0 and self._tiled_supported and self.scatter_mode == 0 and not self._use_dev_scalars
                and self._tbl.numel() * 24 >= self.adam_tiled_min_bytes)

    def _tiled_table(self):
        if self._tiled is None:
            if not 2 <= self.adam_window <= _lib.ADAM_RING - 2:
                raise ValueError("adam_window must be 0 (off) or 2..%d" % (_lib.ADAM_RING - 2))
            row_step = torch.zeros((self._tbl.shape[0],), dtype=torch.int32, device=self.device)
            ring = torch.zeros((_lib.ADAM_RING + 1,), dtype=torch.float32, device=self.device)
            T = _lib.AdamTable(p=_ptr(self._tbl), m=_ptr(self._tbl_m), v=_ptr(self._tbl_v), g=_ptr(self.table_g),
                               n_rows=self._tbl.shape[0], D=self._tbl.shape[1], row_flags=_ptr(self.table_flags),
                               row_step=_ptr(row_step), alpha_ring=_ptr(ring), beta1=ADAM_B1, beta2=ADAM_B2, eps=ADAM_EPS,
                               id_status=_ptr(self._id_status) if self._guard_on else None,
                               skipped_steps=C.c_void_p(self._id_status.data_ptr() + 4) if self._guard_on else None)
            self._tiled = (row_step, ring, T)
            self._tiled_ready = False
        return self._tiled

    def _join_sweep(self, cur, rows_untouched=False):
        """rows_untouched: the caller launches nothing that replays rows on `cur` (the batch's rows were caught up a step ahead);
        a slice the one-call step queued LAST on the side stream then needs no wait: what follows on the side stream queues
        behind it, the forward pass reads rows it skips, the touched-row update publishes counts it cannot mistake"""
        if self._ev_sweep is not None and rows_untouched and self._sweep_on_side and str(self.adam_sweep_at) != "f1":
            pass            # (adam_sweep_at "f1" puts this step's slice on a stream of its own: it would not queue behind that one)
        elif self._ev_sweep is not None:
            cur.wait_event(self._ev_sweep)
            self._ev_sweep = None
        if self._ahead is not None:          # (a look-ahead catch-up replays rows on the side stream: nothing else may meanwhile)
            cur.wait_event(self._ahead[1])

    def _catchup(self, db, sweep):
        """Before a forward: the rows this batch reads are brought up to self.step (score_adam_catchup_ids); in a
        training step the window's slice of the table follows on its own stream, beside the step."""
        ah, self._ahead = self._ahead, None
        if ah is not None:
            self._cur().wait_event(ah[1])
            if ah[0] is db and db.flat is not None:
                # apply_adam(next_batch=db) of the previous step has brought these rows up to date through that step
                # (score_adam_catchup_ids_through, beside its weight-gradient products): nothing to replay here
                ev = self.catchup_events
                if ev:
                    ev[0].record(); ev[1].record()
                self._catchup_ids([], sweep)
                return
        self._catchup_ids([db.flat] if db.flat is not None else list(db.tensors[:6]), sweep)

    def _catchup_ids(self, spans, sweep, inline_sweep=False):
        """spans: int32 device tensors of row ids (values outside the table are ignored); [] = the rows are up to date already"""
        ev = self.catchup_events
        _, _, T = self._tiled_table()     # (created here, on the main stream, well before the side stream first uses it)
        if not self._adam_dirty:
            if ev and spans:
                ev[0].record(); ev[1].record()
            return
        cur = self._cur()
        self._join_sweep(cur, rows_untouched=not spans and not self._flags_marked and not inline_sweep)
        if self._flags_marked:
            self._drop_row_marks()        # (a backward nobody applied left state-2 marks)
        upto = int(self.step)
        if ev and spans:
            ev[0].record()
        for t in spans:
            _lib.check(self.lib.score_adam_catchup_ids(C.byref(T), _ptr(t), t.numel(), upto, self._stream()),
                       "score_adam_catchup_ids")
        if ev and spans:
            ev[1].record()
        if sweep:
            # the window's slice of the table: must start after the rows above are done (it would take them for lagging
            # ones) and finish before the next step's catch-up; forward_backward starts it on the side stream beside
            # the backward recurrence (inline_sweep: here on the main stream instead -- a row shard's gather, score_amd/dist.py)
            rows, K = self._tbl.shape[0], self.adam_window
            j = (upto + 1) % K
            self._pending_sweep = (rows * j // K, rows * (j + 1) // K, upto, self._rec("sweep_after", cur))
            if inline_sweep:
                self._launch_sweep(cur)

    def _launch_sweep(self, stream, behind=None):
        """behind: an event the slice also has to wait for (the stage boundary it starts at)"""
        if self._pending_sweep is None:
            return
        lo, hi, upto, after = self._pending_sweep
        self._pending_sweep = None
        _, _, T = self._tiled
        cur = self._cur()
        other = stream.cuda_stream != cur.cuda_stream       # (current_stream() returns a new wrapper object every call)
        if behind is not None:
            stream.wait_event(behind)
        if other:
            stream.wait_event(after)
        _lib.check(self.lib.score_adam_catchup_rows(C.byref(T), lo, hi, upto, C.c_void_p(stream.cuda_stream)),
                   "score_adam_catchup_rows")
        if other:
            self._ev_sweep = self._rec("sweep", stream)
            self._sweep_on_side = False

    _look_ahead = True               # score_backward records its stage boundary 4 for apply_adam(next_batch=)

    def _catchup_ahead(self, nxt, lr):
        ev4 = getattr(self, "_b4_recorded", None)
        if (not isinstance(nxt, DeviceBatch) or nxt.flat is None or ev4 is None or not self._tiled_ready
                or self._tiled is None or self._side is None):
            return
        _, _, T = self._tiled
        side = self._side
        # (the side stream has run this step's window slice before -- it was started at boundary 2 --, so the two replays
        #  never meet on a row; rows in state 2 are skipped here and updated by score_adam_touched on the main stream)
        side.wait_event(ev4)
        if self._ev_sweep is not None:       # (a window slice on a stream of its own: behind it)
            side.wait_event(self._ev_sweep)
        _lib.check(self.lib.score_adam_catchup_ids_through(C.byref(T), _ptr(nxt.flat), nxt.flat.numel(), int(self.step) + 1,
                                                           self._alpha(lr), C.c_void_p(side.cuda_stream)),
                   "score_adam_catchup_ids_through")
        self._ahead = (nxt, self._rec("ahead", side), 0)
        self._b4_recorded = None

    def _plan_ahead(self, nxt):
        """The NEXT batch's index plan (the occurrence sort of the row scatter: it depends on the ids only), on the side stream
        behind THIS step's row scatter -- the last reader of the plan's buffers.  The ids it sees are not reported (the sticky
        status word guards this step's optimizer kernels, which run meanwhile): the forward pass of the batch reports them."""
        ev4 = getattr(self, "_b4_any", None)
        if (not isinstance(nxt, DeviceBatch) or nxt.flat is None or ev4 is None or self._side is None or self.scatter_mode != 0
                or self._graph_on or self._use_dev_scalars or (self._tiled_on() and self.adam_touched_list)):
            return
        lay, ws = self._workspace(nxt.B)
        st = self._st_ahead
        if st is None:
            st = self._st_ahead = _lib.State()
        st.table = self._tbl.data_ptr(); st.n_table_rows = self._tbl.shape[0]; st.w = self._w.data_ptr()
        st.workspace = ws.data_ptr(); st.workspace_bytes = ws.numel() * 4
        st.scatter_mode = 0; st.gemm_mode = int(self.gemm_mode); st.debug_flags = int(self.debug_flags)
        st.context = self._ctx
        st.id_status = None
        self._side.wait_event(ev4)
        _lib.check(self.lib.score_index_plan(C.byref(self.cfg), C.byref(st), C.byref(nxt.struct), 1, 0, self._side_handle),
                   "score_index_plan")
        self._plan_ready = (nxt, self._rec("plan_ahead", self._side), ws.data_ptr(), nxt.active_slices, 0)

    def _adam_table_tiled(self, lr, dense=None):
        """ApplyAdam of step self.step + 1 on the rows that have a gradient; every other live row owes it.
        (On the main stream: behind the row-gradient event on the side stream, beside the weight-gradient products,
        it slowed those by what it saved -- bwd_weight_grads 0.180 -> 0.246 ms, profiles/r02_probes.md.)
        dense = (reg_lambda,): the flat dense variables' ApplyAdam in the same launch (score_adam_touched_and_dense);
        returns True if it did that."""
        row_step, ring, T = self._tiled_table()
        cur = self._cur()
        # (round 5: NOT behind this step's window slice or the look-ahead catch-up any more -- they run on the side stream, on rows
        #  this update does not touch, and the kernels publish / read a row's step count so that neither can take a row of this
        #  batch for a lagging one (csrc/adam_tiled.hip tiled_publish_applied).  At the CCMR shape the launch stream stood ~95 us
        #  per step waiting for slice -> mark -> catch-up in front of this launch.  The next forward pass still waits for both.)
        if not self._tiled_ready:
            self._join_sweep(cur)
            # (only reached with no row lagging: _flush_adam ran, or nothing tiled has happened yet)
            row_step.fill_(int(self.step))
            self._tiled_ready = True
        rl, self._row_list = getattr(self, "_row_list", None), None
        did_dense = False
        if rl is not None:
            lay, ws = rl          # (the plan's buffers: its side-stream work is behind the event score_backward waited for)
            rows = ws[lay.plan_unique_rows:]
            meta = ws[lay.plan_meta:]
            _lib.check(self.lib.score_adam_touched_rows(C.byref(T), _ptr(rows), _ptr(meta), int(lay.n_occurrences) + 1,
                                                        int(self.step) + 1, self._alpha(lr), self._stream()),
                       "score_adam_touched_rows")
        elif dense is not None and not self._use_dev_scalars:
            _lib.check(self.lib.score_adam_touched_and_dense(
                C.byref(T), int(self.step) + 1, self._alpha(lr), _ptr(self.w), _ptr(self.w_m), _ptr(self.w_v), _ptr(self._w_g),
                self.n_w, self.n_reg, float(dense[0]),
                C.c_void_p(self._id_status.data_ptr() + 4) if self._guard_on else None, self._stream()),
                "score_adam_touched_and_dense")
            did_dense = True
        else:
            _lib.check(self.lib.score_adam_touched(C.byref(T), int(self.step) + 1, self._alpha(lr), self._stream()),
                       "score_adam_touched")
        self._row_grads = False
        self._flags_marked = False
        self._adam_dirty = True
        return did_dense

    def _flush_adam(self):
        """Every live row up to self.step: what any reader of table / table_m / table_v other than the training
        step itself sees first (no-op unless tiled steps have run since the last flush)."""
        if not self._adam_dirty:
            return
        self._adam_dirty = False
        self._pending_sweep = None
        _, ring, T = self._tiled
        cur = self._cur()
        self._join_sweep(cur)
        # (rows in state 2 -- a gradient not applied yet -- are left alone, marks and all: they were brought up to
        #  date before the forward that produced the gradient, and the update that follows still needs the marks)
        _lib.check(self.lib.score_adam_catchup_rows(C.byref(T), 0, self._tbl.shape[0], int(self.step), self._stream()),
                   "score_adam_catchup_rows")
        if int(ring[_lib.ADAM_RING].view(torch.int32).item()) != 0:
            raise RuntimeError("time-tiled Adam: a row lagged more steps than the alpha ring holds")
        if self._guard_on:
            self.check_ids()     # (a set word suppressed the catch-up above: the caller must not read a stale table unawares)

    def adam_table(self, lr):
        """ApplyAdam over the table (shard) on the current stream: needs the row gradients only."""
        a = self._alpha(lr)
        s = self._stream()
        self._flush_adam()
        self._tiled_ready = False           # a sweep moves every row: row_step no longer describes the table
        if self._row_grads:
            if self._use_dev_scalars:
                rc = self.lib.score_adam_rows_dev(_ptr(self._tbl), _ptr(self._tbl_m), _ptr(self._tbl_v),
                                                  _ptr(self.table_g), self._tbl.shape[0], self._tbl.shape[1],
                                                  _ptr(self.table_flags), _ptr(self._scalars), ADAM_B1, ADAM_B2, ADAM_EPS,
                                                  self._guard_table, s)
            else:
                rc = self.lib.score_adam_rows(_ptr(self._tbl), _ptr(self._tbl_m), _ptr(self._tbl_v),
                                              _ptr(self.table_g), self._tbl.shape[0], self._tbl.shape[1],
                                              _ptr(self.table_flags), a, ADAM_B1, ADAM_B2, ADAM_EPS, self._guard_table, s)
            self._row_grads = False
            self._flags_marked = False
        else:
            rc = self.lib.score_adam(_ptr(self._tbl), _ptr(self._tbl_m), _ptr(self._tbl_v), _ptr(self.table_g),
                                     self._tbl.numel(), 0, 0.0, a, ADAM_B1, ADAM_B2, ADAM_EPS, self._guard_table, s)
            self.table_flags.fill_(1)       # dense sweep: any row may carry moments now
            self._flags_marked = False
        _lib.check(rc, "score_adam(table)")

    def adam_dense(self, lr, reg_lambda, stream=None):
        """ApplyAdam over the flat dense variables (L2 term folded in) on the current stream.
        stream (apply_adam only): a side stream the caller has ordered behind the dense gradient's finishers."""
        if stream is not None:
            rc = self.lib.score_adam(_ptr(self._w), _ptr(self._w_m), _ptr(self._w_v), _ptr(self._w_g), self.n_w,
                                     self.n_reg, float(reg_lambda), self._alpha(lr), ADAM_B1, ADAM_B2, ADAM_EPS,
                                     self._guard_dense, C.c_void_p(stream.cuda_stream))
            _lib.check(rc, "score_adam(dense)")
            return
        if self._use_dev_scalars:
            rc = self.lib.score_adam_dev(_ptr(self.w), _ptr(self.w_m), _ptr(self.w_v), _ptr(self.w_g), self.n_w,
                                         self.n_reg, float(reg_lambda), _ptr(self._scalars), ADAM_B1, ADAM_B2, ADAM_EPS,
                                         self._guard_dense, self._stream())
        else:
            rc = self.lib.score_adam(_ptr(self.w), _ptr(self.w_m), _ptr(self.w_v), _ptr(self.w_g), self.n_w,
                                     self.n_reg, float(reg_lambda), self._alpha(lr), ADAM_B1, ADAM_B2, ADAM_EPS,
                                     self._guard_dense, self._stream())
        _lib.check(rc, "score_adam(dense)")

    def adam_advance(self):
        """beta1^t, beta2^t and the step counter: once per step, after both halves."""
        self.beta1_power = np.float32(self.beta1_power * np.float32(ADAM_B1))
        self.beta2_power = np.float32(self.beta2_power * np.float32(ADAM_B2))
        self.step += 1

    # "auto" = two alternating plan buffers below this many occurrences per batch.  With two, the next batch's sort (~100 us) runs
    # beside this step's passes instead of behind its scatter: at few occurrences per batch that sort is the longest cycle of the
    # step, at many the launch stream's chain is and the early sort only slows the kernels on it.  One box, interleaved, 2,000
    # steps, one / two buffers (profiles/r06_probes.md): Taobao default (96.6 K occurrences) 0.2183 / 0.2200 vs 0.1932 / 0.1928 ms,
    # cfg-2 (181 K) 0.1637 / 0.1639 vs 0.1643 / 0.1635, Tmall default (309 K) 0.1891 / 0.1880 vs 0.1927 / 0.1913, CCMR default
    # (961 K) 0.3451 / 0.3504 vs 0.3584 / 0.3562.  (Until round 5 "auto" TIMED both inside the first ~160 steps of a run: the
    # choice could differ from run to run and rank to rank, and the tuning fell into short measurements.)
    TWO_BUFFERS_BELOW = 150000

    def _two_buffers(self, nxt):
        v = self.plan_two_workspaces
        if v is True or v is False:
            return v
        c = self.cfg
        slices = int(nxt.active_slices) or int(c.max_time_len)
        f = int(c.user_fnum) + int(c.item_fnum)
        return int(nxt.B) * (2 * slices * int(c.obj_per_time_slice) * f + f) < self.TWO_BUFFERS_BELOW

    def _plan_buffer(self, B, which):
        """address of plan buffer 0 / 1 of a batch size: the step's workspace, and a second one of the same layout that only ever
        holds index plans (the one-call step alternates the two: score_state_t.plan_workspace)"""
        e = self._pb_cache.get(B)
        if e is None:         # (emptied by _workspace whenever a workspace is created or evicted)
            w0, w1 = self._workspace(B, 0)[1].data_ptr(), self._workspace(B, 1)[1].data_ptr()
            e = self._pb_cache[B] = (w0, w1)
        return e[which]

    def _ensure_ev(self, attr):
        ev = getattr(self, attr)
        if ev is None:
            ev = self._ev_new(host=attr == "_ev_loss")       # (the host reads the loss behind _ev_loss)
            ev.record(self._cur())              # materialise the hipEvent_t
            setattr(self, attr, ev)
        return ev

    def _train_step_fast(self, db, lr, reg_lambda, keep_prob, nxt):
        """The steady-state step of the per-sample form as ONE call into the library (score_train_step, csrc/step.hip): the same
        entry points with the same arguments on the same streams and events as forward_backward + apply_adam below make one by
        one -- what is saved is this interpreter's share of the step (tools/host_calls.py: ~85 of ~190 us at the Tmall default
        shape, which is bound by the host).  Returns None when the step is not that steady state (first steps, another batch than
        the one announced, stage events, an evaluation in between, ...): the caller then takes the call-by-call path, and the two
        can alternate step by step (tests/test_gpu_persample.py)."""
        if (not self.fast_step or self._graph_on or self._use_dev_scalars or self.scatter_mode != 0 or self.bwd_events
                or self.catchup_events or int(self.debug_flags) or str(self.adam_sweep_at) not in ("2", "auto") or self.adam_touched_list
                or self._tiled is None or not self._tiled_ready or not self._adam_dirty or self._flags_marked or self._row_grads
                or self._pending_sweep is not None or self._side is None or not isinstance(db, DeviceBatch) or db.flat is None
                or not self._look_ahead or not self._tiled_on()):
            return None
        ah, pr = self._ahead, self._plan_ready
        if ah is None or ah[0] is not db or pr is None or pr[0] is not db or pr[3] != db.active_slices:
            return None
        if nxt is not None and (not isinstance(nxt, DeviceBatch) or nxt.flat is None or not self.plan_ahead):
            return None
        if not self.persample_form(db.B, db.active_slices):
            return None
        # two plan buffers per batch size (the step's workspace and a second one of the same layout that only ever holds plans),
        # alternating: the next batch's plan goes into the one this batch's plan is NOT in, so its sort needs nothing of this step
        # (score_train_step_t.ev_plan_next); everything else of every step stays in the one workspace
        lay, ws = self._workspace(db.B)
        slot = 0 if pr[2] == self._plan_buffer(db.B, 0) else 1 if pr[2] == self._plan_buffer(db.B, 1) else -1
        if slot < 0:
            return None
        cur = self._cur()
        if self._inline_on or (self._train_stream is not None and self._train_stream.cuda_stream != cur.cuda_stream):
            return None
        self._join_dense()
        self._join_grads()
        st = self._state(ws)
        st.plan_workspace = pr[2] if slot == 1 else None
        row_step, ring, T = self._tiled
        p = self._step_args
        if p is None:
            p = self._step_args = _lib.TrainStep()
            p.table = C.addressof(T)
            p.n_w, p.n_reg = self.n_w, self.n_reg
            self._step_T = T
        elif self._step_T is not T:
            p.table = C.addressof(T)
            self._step_T = T
        # (the buffers re-read every call: set_params replaces them; the side stream when it has been replaced -- the inline mode
        #  switched on and off --; the events are created once and never replaced)
        p.w, p.w_m, p.w_v, p.w_g = self._w.data_ptr(), self._w_m.data_ptr(), self._w_v.data_ptr(), self._w_g.data_ptr()
        if self._step_side is not self._side:
            self._step_side = self._side
            p.skipped = self._id_status.data_ptr() + 4 if self._guard_on else None
            p.side_stream = self._side.cuda_stream
            p.ev_stage2 = None
            p.ev_b4 = self._ensure_ev("_ev_b4").cuda_event
            p.ev_grads = self._ensure_ev("_ev_grads").cuda_event
            p.ev_loss = self._ensure_ev("_ev_loss").cuda_event
        fe = self._event_array(self.fwd_events) if self.fwd_events else None      # (a caller timing the forward pass: bench.py's roofline)
        p.fwd_stage_events = C.cast(fe, C.c_void_p) if fe is not None else None
        # (the previous step's window slice: waited for only if it ran somewhere else than in front of the look-ahead catch-up of
        #  this batch's rows on the side stream -- i.e. after a call-by-call step with its slice on another stream; the one-call
        #  step queues its slice LAST on the side stream, and the next one's work there queues behind it)
        ev_sweep = self._ev_sweep
        p.wait_sweep = 1 if (ev_sweep is not None and not self._sweep_on_side) else 0
        if ev_sweep is None:
            ev_sweep = self._evs.get("sweep") or self._rec("sweep", cur)
        p.ev_sweep = ev_sweep.cuda_event
        p.wait_ahead = 1
        p.ev_ahead = ah[1].cuda_event
        p.ev_plan = pr[1].cuda_event
        p.reg_lambda, p.keep_prob, p.alpha = float(reg_lambda), float(keep_prob), self._alpha(lr)
        upto = int(self.step)
        p.step = upto + 1
        p.drop_seed = (self._drop_seed * 0x9E3779B1 + upto * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
        rows, K = self._tbl.shape[0], self.adam_window
        j = (upto + 1) % K
        p.slice_lo, p.slice_hi, p.slice_upto = rows * j // K, rows * (j + 1) // K, upto
        ev_plan_next = None
        if nxt is not None:
            lay2, ws2 = self._workspace(nxt.B, (1 - slot) if (self._two_buffers(nxt) and nxt.B == db.B) else 0)
            p.next_batch, p.next_ids, p.n_next_ids = C.addressof(nxt.struct), nxt.flat.data_ptr(), nxt.flat.numel()
            p.next_workspace, p.next_workspace_bytes = ws2.data_ptr(), ws2.numel() * 4
            if ws2.data_ptr() != pr[2]:
                # (an event of its own: this step's scatter still has to wait for THIS batch's plan event)
                ev_plan_next = self._plan_events[1] if pr[1] is self._plan_events[0] else self._plan_events[0]
                if ev_plan_next is None:
                    ev_plan_next = self._ev_new()
                    ev_plan_next.record(cur)
                    self._plan_events[0 if self._plan_events[0] is None else 1] = ev_plan_next
            p.ev_plan_next = ev_plan_next.cuda_event if ev_plan_next is not None else None
            if ev_plan_next is not None and self._plan_stream is None:
                self._plan_stream = torch.cuda.Stream(device=self.device)
            p.plan_stream = self._plan_stream.cuda_stream if ev_plan_next is not None else None
        else:
            p.next_batch = None
            p.ev_plan_next = None
            p.plan_stream = None
        el = self._early_loss
        if el is not None:
            if el["host"] is None:
                el["host"] = torch.zeros((4,), dtype=torch.float32).pin_memory()
            p.loss_host = el["host"].data_ptr()
            el["event"] = self._ev_loss
        else:
            p.loss_host = None
        self._ahead = None
        self._plan_ready = None
        _lib.check(self.lib.score_train_step(C.byref(self.cfg), C.byref(st), C.byref(db.struct), C.byref(p), self._stream()),
                   "score_train_step")
        # the bookkeeping forward_backward + apply_adam leave behind
        self._ps_last, self._train_stream = True, cur
        self._keep = (None, None)
        self._ev_sweep = ev_sweep if p.slice_hi > p.slice_lo else None
        self._sweep_on_side = True
        self._plan_done = pr[1]
        self._b4_recorded, self._b4_any = None, self._ev_b4
        self._row_grads = self._flags_marked = False
        self._adam_dirty = True
        self._row_list = None
        self._grads_pending = None
        if nxt is not None:
            self._ahead = (nxt, ah[1], 0)
            self._plan_ready = (nxt, ev_plan_next if ev_plan_next is not None else pr[1], ws2.data_ptr(), nxt.active_slices, 0)
        self.adam_advance()
        return ws[lay.loss]

    def train_async(self, batch_data, lr, reg_lambda, keep_prob=0.8, dropout_masks=None, next_batch=None):
        """One training step; returns the loss as a 0-d device tensor (no host sync).  next_batch: see apply_adam."""
        if self._graph_on and dropout_masks is None and self.scatter_mode == 0 and not self.fwd_events:
            return self._train_captured(batch_data, lr, reg_lambda, keep_prob)
        if dropout_masks is None and self._ahead is not None and self._ahead[0] is batch_data:
            with self._Pin(self):
                loss = self._train_step_fast(batch_data, lr, reg_lambda, keep_prob, next_batch)
            if loss is not None:
                return loss
        with self._Pin(self):
            lay, ws = self.forward_backward(batch_data, reg_lambda, keep_prob, dropout_masks)
            self.apply_adam(lr, reg_lambda, next_batch)
        return ws[lay.loss]

    # ------------------------------------------------------------------ captured step (hipGraph)
    def enable_graph(self, on=True):
        """Replay the training step as ONE captured hipGraph per (batch size, active slices, reg_lambda, keep_prob):
        small shapes (the reference's own B = 100 / 200, D = 16, H = 32) are launch-bound -- ~60 launches of a few
        microseconds of work each.  The first step of a shape runs eagerly, the second is captured, later ones are
        replays of it: the batch is copied into the capture's static batch (one device-to-device copy), alpha and
        the dropout seed are rewritten in device memory (score_step_scalars_t).  Same kernels, same arguments, same
        order as the eager step: results are bit-identical (tests/test_gpu_graph.py)."""
        self._graph_on = bool(on)
        if not on:
            self._graphs = {}
            self._use_dev_scalars = False

    SCALAR_SLOTS = 16

    def _write_step_scalars(self, lr):
        if self._scalars_ring is None:
            self._scalars_ring = [[torch.zeros((4,), dtype=torch.int32).pin_memory(), None] for _ in range(self.SCALAR_SLOTS)]
        slot = self._scalars_ring[self._scalars_slot]
        self._scalars_slot = (self._scalars_slot + 1) % self.SCALAR_SLOTS
        if slot[1] is not None:
            slot[1].synchronize()          # the copy that last read this slot (SCALAR_SLOTS steps ago) has run
        h = slot[0]
        h[0] = int(np.float32(self._alpha(lr)).view(np.int32))
        seed = (self._drop_seed * 0x9E3779B1 + self.step * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
        lo, hi = seed & 0xFFFFFFFF, seed >> 32
        h[1] = 0
        h[2] = lo - (1 << 32) if lo >= (1 << 31) else lo
        h[3] = hi - (1 << 32) if hi >= (1 << 31) else hi
        self._scalars.copy_(h, non_blocking=True)
        slot[1] = torch.cuda.current_stream(self.device).record_event()

    def _train_captured(self, batch_data, lr, reg_lambda, keep_prob):
        db = self.device_batch(batch_data)
        # everything the launch sequence bakes in by value (pointers are stable: the table, the flat parameter
        # buffers and the workspace of a batch size never move)
        key = (db.B, db.active_slices, float(reg_lambda), float(keep_prob), int(self.global_batch), int(self.gemm_mode),
               int(self.debug_flags))
        ent = self._graphs.get(key)
        self._join_dense()               # (eager steps before: nothing of theirs may be pending on a side stream when a capture begins)
        self._join_grads()
        self._use_dev_scalars = True
        try:
            self._write_step_scalars(lr)
            if ent is None or ent == "warm":
                # eager (first: allocates the workspace, creates streams / events; second: settles the caching allocator)
                lay, ws = self.forward_backward(db, reg_lambda, keep_prob, None)
                self.apply_adam(lr, reg_lambda)
                self._graphs[key] = "warm" if ent is None else "ready"
                return ws[lay.loss]
            if ent == "ready":
                self._flush_adam()
                self._tiled_ready = False
                static = DeviceBatch.empty(self, db.B, db.active_slices)
                static.flat.copy_(db.flat)
                lay, ws = self._workspace(db.B)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                step, b1p, b2p = self.step, self.beta1_power, self.beta2_power
                # (no cyclic garbage collection inside the capture: another model's __del__ -- it synchronises its streams and
                #  destroys its context -- is not a legal call while a stream of the device is capturing; seen once as
                #  hipErrorStreamCaptureInvalidated in a long test session.  torch.cuda.graph collects right before it begins.)
                import gc
                gc_on = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g):
                        self.forward_backward(static, reg_lambda, keep_prob, None)
                        self.apply_adam(lr, reg_lambda)
                finally:
                    if gc_on:
                        gc.enable()
                # (capturing executed nothing: undo the host-side bookkeeping of the traced call)
                self.step, self.beta1_power, self.beta2_power = step, b1p, b2p
                ent = self._graphs[key] = (g, static, lay, ws)
            g, static, lay, ws = ent
            # the capture bakes the per-step SWEEP in: every update the time-tiled optimizer still owes (an eager step
            # in between -- explicit dropout masks, stage events -- may have run it) is applied first, and row_step no
            # longer describes the table after the replayed sweep
            self._flush_adam()
            self._tiled_ready = False
            static.flat.copy_(db.flat, non_blocking=True)
            g.replay()
            self.adam_advance()
            return ws[lay.loss]
        finally:
            self._use_dev_scalars = False

    # ------------------------------------------------------------------ reference interface
    def train(self, sess, batch_data, lr, reg_lambda, keep_prob=0.8, dropout_masks=None, next_batch=None):
        """loss = model.train(sess, batch_data, lr, reg_lambda) (score.py:101-116).  The loss is read back every step, as the
        reference's sess.run returns it -- from a copy taken right behind the forward pass on a stream of its own, so the
        read-back does not drain the queue: the backward pass and the optimizer of this step run while the host prepares the
        next call (every later use of the model is ordered behind them on the stream)."""
        # (not for captured steps: the copy's stream would be unjoined work inside the capture)
        self._early_loss = None if self._graph_on else self._early_loss_state
        self._early_loss_state["event"] = None
        try:
            dev_loss = self.train_async(batch_data, lr, reg_lambda, keep_prob, dropout_masks, next_batch)
        finally:
            self._early_loss = None
        ev = self._early_loss_state["event"]
        if ev is not None:
            ev.synchronize()
            loss = float(self._early_loss_state["host"][0])
        else:                       # (a captured step: the loss comes at the end of the replay)
            loss = float(dev_loss.item())
        if loss != loss:
            self.check_ids()
        return loss

    def check_ids(self):
        """Raises ValueError if any batch fed since the last call held a feature id outside [0, feature_size) --
        where tf.nn.embedding_lookup raises InvalidArgumentError (score.py:51-66).  The kernels that read the ids
        report into a sticky device word (score_state_t.id_status) and treat such an id as the dummy row 0, so
        nothing is ever read or written out of bounds; the loss of such a step is NaN, which is how train() / eval()
        learn of it without an extra read-back.  As in TF, where the exception leaves sess.run before any assign op has run
        (score.py:101-116), NO variable has been updated by the offending step: the optimizer kernels read the same
        word when they execute and apply nothing while it is set (score_guard_t); the device counts the steps it
        suppressed -- with train_async the host may have queued several more behind the offending one -- and this
        call takes them off the step count and the beta powers again.  Parameters, Adam slots, step and beta powers
        are then bit for bit what they were before the offending call.  Callers of train_async / eval_async call
        this at their own sync points."""
        self._join_dense()        # (the dense variables' update may be on the side stream: it is what counts a suppressed step)
        bits, skipped = [int(x) for x in self._id_status.tolist()]
        if bits:
            self._id_status.zero_()
            if skipped:
                self._rollback_steps(skipped)
            self._pending_sweep = None          # (a window slice scheduled for a step that was not applied)
            if self._ahead is not None:         # (a look-ahead catch-up that the set word suppressed)
                self._cur().wait_event(self._ahead[1])
                self._ahead = None
            self._adam_dirty = self._tiled is not None and self._tiled_ready     # (a suppressed flush left rows behind)
            self._flags_marked = True           # state-2 marks of the suppressed steps: gone before the next backward
            self._drop_row_marks()
            names = ["batch_data[%d] (%s)" % (i, BATCH_FIELDS[i]) for i in range(6) if bits >> i & 1]
            raise ValueError("feature id outside [0, %d) in %s (tf.nn.embedding_lookup would raise: score.py:51-66); "
                             "no variable was updated%s"
                             % (int(self.cfg.feature_size), ", ".join(names),
                                "" if skipped <= 1 else " by that step or the %d queued behind it" % (skipped - 1)))

    def _rollback_steps(self, k):
        """the host's step count and beta powers, k optimizer steps back (the device applied none of them)"""
        self.step = max(0, int(self.step) - int(k))
        f = np.float32
        # beta^(step+1) as adam_advance computes it: a chain of fp32 products (not a closed form: the bits must match)
        self.beta1_power = f(np.multiply.accumulate(np.full((self.step + 1,), ADAM_B1, dtype=f))[-1])
        self.beta2_power = f(np.multiply.accumulate(np.full((self.step + 1,), ADAM_B2, dtype=f))[-1])

    def eval_async(self, batch_data, reg_lambda):
        """eval without the host round trip: (y_pred [B] device view of the workspace -- copy it before the next
        forward --, labels [B] device int32, loss 0-d device tensor)."""
        db = self.device_batch(batch_data)
        with self._Pin(self):
            lay, ws, _ = self._forward(db, reg_lambda, 1.0, None)
        return ws[lay.y_pred:lay.y_pred + db.B], db.tensors[6], ws[lay.loss]

    def eval(self, sess, batch_data, reg_lambda):
        db = self.device_batch(batch_data)
        with self._Pin(self):
            lay, ws, _ = self._forward(db, reg_lambda, 1.0, None)
        B = db.B
        pred = ws[lay.y_pred:lay.y_pred + B].cpu().numpy()
        label = db.tensors[6].cpu().numpy()
        loss = float(ws[lay.loss].item())
        if loss != loss:
            self.check_ids()
        return pred.reshape([-1, ]).tolist(), label.reshape([-1, ]).tolist(), loss

    def save(self, sess, path):
        """All global variables incl. the Adam slots, under the TF variable names."""
        if self._guard_on:
            # an id outside the table fed through train_async / eval_async and never checked: the optimizer has applied nothing
            # since, while the host's step count and beta powers went on -- such a checkpoint would hold global_step / beta
            # powers ahead of the variables.  Raises (and takes the suppressed steps off the count) instead of writing it.
            self.check_ids()
        blob = {"emb_mtx": self._table_host()}
        for e in self.entries:
            blob[e[0]] = self._view(self.w, e).cpu().numpy().copy()
        tm, tv = self.table_m.cpu().numpy(), self.table_v.cpu().numpy()
        blob["emb_mtx/Adam"], blob["emb_mtx/Adam_1"] = tm, tv
        for e in self.entries:
            blob[e[0] + "/Adam"] = self._view(self.w_m, e).cpu().numpy()
            blob[e[0] + "/Adam_1"] = self._view(self.w_v, e).cpu().numpy()
        blob["beta1_power"] = self.beta1_power
        blob["beta2_power"] = self.beta2_power
        blob["global_step"] = np.int64(self.step)
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(path + ".npz", "wb") as f:
            np.savez(f, **blob)

    def restore(self, sess, path):
        z = np.load(path + ".npz")
        self._table_load(z["emb_mtx"])
        self._set_dense({e[0]: z[e[0]] for e in self.entries})
        self.table_m.copy_(torch.from_numpy(z["emb_mtx/Adam"]))
        self.table_v.copy_(torch.from_numpy(z["emb_mtx/Adam_1"]))
        for e in self.entries:
            self._view(self.w_m, e).copy_(torch.from_numpy(z[e[0] + "/Adam"]).view_as(self._view(self.w_m, e)))
            self._view(self.w_v, e).copy_(torch.from_numpy(z[e[0] + "/Adam_1"]).view_as(self._view(self.w_v, e)))
        self.refresh_row_flags()
        self.beta1_power = np.float32(z["beta1_power"])
        self.beta2_power = np.float32(z["beta2_power"])
        self.step = int(z["global_step"])
        print('model restored from {}'.format(path))


class SCORE(SCOREBASE):
    model_type = "SCORE"


class RIA(SCOREBASE):
    model_type = "RIA"


class RCA(SCOREBASE):
    model_type = "RCA"


class SCORE_USER(SCOREBASE):
    model_type = "SCORE_USER"


class SCORE_ITEM(SCOREBASE):
    model_type = "SCORE_ITEM"


class RRN(SCOREBASE):
    """slice_models/slice_model.py:155-174: the summed 1-hop sets feed the two GRUs, final states + targets
    feed the same head; same constructor and train/eval/save/restore (SliceBaseModel, :11-152)."""
    model_type = "RRN"


MODELS = {"SCORE": SCORE, "RIA": RIA, "RCA": RCA, "SCORE_USER": SCORE_USER, "SCORE_ITEM": SCORE_ITEM, "RRN": RRN}
