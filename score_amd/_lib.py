"""ctypes binding of libscore_hip.so (include/score_hip.h).

The HIP library is the product path: there is NO CPU or PyTorch fallback.  If the
shared object is missing it is built in-tree with hipcc; if that fails, importing
this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libscore_hip.so")

MODEL_TYPES = {"SCORE": 0, "RIA": 1, "RCA": 2, "SCORE_USER": 3, "SCORE_ITEM": 4, "RRN": 5}

c_f = C.c_void_p   # device float*
c_i = C.c_void_p   # device int32*


class Config(C.Structure):
    _fields_ = [("feature_size", C.c_int64), ("eb_dim", C.c_int32), ("hidden_size", C.c_int32),
                ("max_time_len", C.c_int32), ("obj_per_time_slice", C.c_int32),
                ("user_fnum", C.c_int32), ("item_fnum", C.c_int32), ("model_type", C.c_int32)]


class ParamEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("regularised", C.c_int32), ("init", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("user_1hop", c_i), ("user_2hop", c_i), ("item_1hop", c_i), ("item_2hop", c_i),
                ("target_user", c_i), ("target_item", c_i), ("label", c_i), ("length", c_i),
                ("B", C.c_int32), ("active_slices", C.c_int32)]


class Workspace(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("total_bytes", "xside", "atten_info", "rsave", "query", "head_inp", "att_score",
                 "logit", "y_pred", "loss", "gru_out", "gru_final", "n_occurrences", "plan_meta",
                 "plan_unique_rows")] + [("plan_remap", C.c_int64 * 6)]


class State(C.Structure):
    _fields_ = [("table", c_f), ("n_table_rows", C.c_int64), ("w", c_f), ("workspace", c_f),
                ("workspace_bytes", C.c_int64), ("scatter_mode", C.c_int32), ("global_batch", C.c_int32),
                ("gemm_mode", C.c_int32), ("debug_flags", C.c_int32), ("row_flags", C.c_void_p),
                ("gather_done_event", C.c_void_p), ("plan_done_event", C.c_void_p), ("step_scalars", C.c_void_p),
                ("context", C.c_void_p), ("id_status", C.c_void_p), ("grads_done_event", C.c_void_p),
                ("loss_done_event", C.c_void_p), ("loss_host", C.c_void_p), ("plan_workspace", C.c_void_p),
                ("reserved4", C.c_int32), ("reserved3", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("user_off1", "user_nbr1", "user_off2", "user_nbr2", "item_off1", "item_nbr1",
                                           "item_off2", "item_nbr2", "user_rows", "item_rows")] + \
               [(n, C.c_int32) for n in ("n_users", "n_items", "time_slice_num", "user_fnum", "item_fnum", "sample_mode")] + \
               [("user_deg2", C.c_void_p), ("item_deg2", C.c_void_p)]


class BatchOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("user_1hop", "user_2hop", "item_1hop", "item_2hop", "target_user",
                                           "target_item", "label", "length")]


class AdamTable(C.Structure):
    """score_adam_table_t"""
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("n_rows", C.c_int64),
                ("D", C.c_int32), ("reserved", C.c_int32), ("row_flags", C.c_void_p), ("row_step", C.c_void_p),
                ("alpha_ring", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("reserved2", C.c_int32), ("id_status", C.c_void_p), ("skipped_steps", C.c_void_p)]


class TrainStep(C.Structure):
    """score_train_step_t"""
    _fields_ = [("table", C.c_void_p), ("w", C.c_void_p), ("w_m", C.c_void_p), ("w_v", C.c_void_p), ("w_g", C.c_void_p),
                ("n_w", C.c_int64), ("n_reg", C.c_int64), ("skipped", C.c_void_p), ("reg_lambda", C.c_float),
                ("keep_prob", C.c_float), ("alpha", C.c_float), ("step", C.c_uint32), ("drop_seed", C.c_uint64),
                ("slice_lo", C.c_int64), ("slice_hi", C.c_int64), ("slice_upto", C.c_uint32), ("wait_ahead", C.c_int32),
                ("wait_sweep", C.c_int32), ("next_batch", C.c_void_p), ("next_ids", C.c_void_p), ("n_next_ids", C.c_int64),
                ("next_workspace", C.c_void_p), ("next_workspace_bytes", C.c_int64), ("loss_host", C.c_void_p),
                ("side_stream", C.c_void_p), ("ev_ahead", C.c_void_p), ("ev_sweep", C.c_void_p), ("ev_plan", C.c_void_p),
                ("ev_stage2", C.c_void_p), ("ev_b4", C.c_void_p), ("ev_grads", C.c_void_p), ("ev_loss", C.c_void_p),
                ("ev_plan_next", C.c_void_p), ("fwd_stage_events", C.c_void_p), ("plan_stream", C.c_void_p)]


class Guard(C.Structure):
    """score_guard_t"""
    _fields_ = [("id_status", C.c_void_p), ("skipped", C.c_void_p)]


ADAM_RING = 64          # SCORE_ADAM_RING

_SIGS = {
    "score_context_create": [C.POINTER(C.c_void_p)],
    "score_context_destroy": [C.c_void_p],
    "score_id_status": [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p],
    "score_stream_copy": [c_f, c_f, C.c_int64, C.c_void_p],
    "score_table_init": [c_f, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p],
    "score_batch_assemble": [C.POINTER(Graph), c_i, c_i, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                             C.c_int32, C.c_uint64, C.POINTER(BatchOut), C.c_void_p],
    "score_param_layout": [C.POINTER(Config), C.POINTER(ParamEntry), C.c_int32, C.POINTER(C.c_int64),
                           C.POINTER(C.c_int64)],
    "score_workspace_layout": [C.POINTER(Config), C.c_int32, C.POINTER(Workspace)],
    "score_workspace_field": [C.POINTER(Config), C.c_int32, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "score_gather_fwd": [c_f, C.c_int64, C.c_int32, c_i, C.c_int64, c_f, C.c_void_p],
    "score_coattn_fwd": [c_f, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i, c_i,
                         c_f, c_f, c_f, c_f, C.c_int32, c_f, C.c_int32, c_f, C.c_int32, c_f, C.c_int32,
                         C.c_void_p],
    "score_coattn_bwd": [c_f, c_f, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i, c_i,
                         c_f, c_f, c_f, C.c_int32, c_f, C.c_int32, c_f, C.c_int32, c_f, c_f, c_f, C.c_int64,
                         C.c_int32, C.c_void_p],
    "score_gemm": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f, C.c_int32, c_f, C.c_int32, c_f, C.c_int32,
                   c_f, C.c_int32, C.c_float, C.c_void_p, C.c_uint64, c_f, C.c_int64, C.c_void_p],
    "score_gemm_panel_images": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, c_f, C.c_int64, C.c_void_p],
    "score_gemm_panel_run": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                             c_f, C.c_int64, C.c_void_p],
    "score_gemm_panel_products": [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, c_f, C.c_int64, C.c_void_p],
    "score_gru_fwd": [C.c_int32, C.c_int32, C.c_int32, c_f, c_f, C.c_int32, c_f, C.c_int32, c_i, c_f, C.c_int32,
                      c_f, c_f, C.c_void_p],
    "score_gru_bwd": [C.c_int32, C.c_int32, C.c_int32, c_f, C.c_int32, c_f, C.c_int32, c_i, c_f, C.c_int32, c_f,
                      c_f, C.c_int32, c_f, c_f, c_f, c_f, C.c_void_p],
    "score_adam": [c_f, c_f, c_f, c_f, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                   C.c_float, C.POINTER(Guard), C.c_void_p],
    "score_adam_dev": [c_f, c_f, c_f, c_f, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_float,
                       C.POINTER(Guard), C.c_void_p],
    "score_adam_rows_dev": [c_f, c_f, c_f, c_f, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                            C.c_float, C.POINTER(Guard), C.c_void_p],
    "score_adam_rows": [c_f, c_f, c_f, c_f, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_float,
                        C.c_float, C.POINTER(Guard), C.c_void_p],
    "score_adam_rows_and_dense": [c_f, c_f, c_f, c_f, C.c_int64, C.c_int32, C.c_void_p, c_f, c_f, c_f, c_f, C.c_int64, C.c_int64,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Guard), C.c_void_p],
    "score_adam_touched": [C.POINTER(AdamTable), C.c_uint32, C.c_float, C.c_void_p],
    "score_adam_touched_and_dense": [C.POINTER(AdamTable), C.c_uint32, C.c_float, c_f, c_f, c_f, c_f, C.c_int64, C.c_int64,
                                     C.c_float, C.c_void_p, C.c_void_p],
    "score_adam_unmark": [C.POINTER(AdamTable), C.c_uint32, C.c_void_p],
    "score_adam_touched_rows": [C.POINTER(AdamTable), c_i, c_i, C.c_int64, C.c_uint32, C.c_float, C.c_void_p],
    "score_adam_catchup_ids": [C.POINTER(AdamTable), c_i, C.c_int64, C.c_uint32, C.c_void_p],
    "score_adam_catchup_ids_through": [C.POINTER(AdamTable), c_i, C.c_int64, C.c_uint32, C.c_float, C.c_void_p],
    "score_adam_catchup_rows": [C.POINTER(AdamTable), C.c_int64, C.c_int64, C.c_uint32, C.c_void_p],
    "score_index_plan": [C.POINTER(Config), C.POINTER(State), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_void_p],
    "score_sort_pairs": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                         C.POINTER(C.c_int32), C.c_void_p],
    "score_sort_pairs_temp_bytes": [C.c_int64],
    "score_segment_sum_rows": [c_i, c_f, C.c_int64, C.c_int32, C.c_int64, c_f, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p],
    "score_segment_sum_scratch_bytes": [C.c_int64, C.c_int32],
    "score_rows_accumulate": [c_i, c_f, C.c_int64, C.c_int32, C.c_int64, c_f, C.c_void_p, C.c_void_p],
    "score_rows_accumulate_multi": [c_i, c_f, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int64, c_f, C.c_void_p, C.c_void_p],
    "score_auc_logloss": [c_f, c_i, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "score_auc_scratch_bytes": [C.c_int64],
    "score_ranking_quality": [c_f, c_i, C.c_int64, C.c_int32, c_f, c_i, c_f, C.c_int64, C.c_void_p],
    "score_persample_form": [C.POINTER(Config), C.POINTER(State), C.c_int32, C.c_int32],
    "score_event_create": [C.POINTER(C.c_void_p)],
    "score_event_destroy": [C.c_void_p],
    "score_event_record": [C.c_void_p, C.c_void_p],
    "score_stream_wait_event": [C.c_void_p, C.c_void_p],
    "score_event_query": [C.c_void_p],
    "score_event_synchronize": [C.c_void_p],
    "score_abi_struct_sizes": [C.POINTER(C.c_int64), C.c_int32],
    "score_train_step": [C.POINTER(Config), C.POINTER(State), C.POINTER(Batch), C.POINTER(TrainStep), C.c_void_p],
    "score_gemm_forms": [C.POINTER(Config), C.POINTER(State), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "score_forward": [C.POINTER(Config), C.POINTER(State), C.POINTER(Batch), C.c_float, C.c_float, C.c_void_p,
                      C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p],
    "score_backward": [C.POINTER(Config), C.POINTER(State), C.POINTER(Batch), C.c_float, c_f, c_f,
                       C.POINTER(C.c_void_p), C.c_void_p],
}

EXPORTS = tuple(_SIGS)
_lib = None


class DevEvent(object):
    """An event that orders DEVICE work only (score_event_create: no system-scope fence at a record), with the methods of
    torch.cuda.Event the host code uses -- record(stream), wait(stream) (what torch.cuda.Stream.wait_event(event) calls),
    query(), synchronize() -- and `.cuda_event`, the hipEvent_t handed to the library.  Not for a host thread that wants to READ
    what was computed in front of it (torch.cuda.Event for that)."""
    __slots__ = ("cuda_event", "_lib")

    def __init__(self):
        lib = load()
        h = C.c_void_p(0)
        check(lib.score_event_create(C.byref(h)), "score_event_create")
        self.cuda_event = h.value
        self._lib = lib

    def record(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        check(self._lib.score_event_record(self.cuda_event, s.cuda_stream), "score_event_record")

    def wait(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        check(self._lib.score_stream_wait_event(s.cuda_stream, self.cuda_event), "score_stream_wait_event")

    def query(self):
        rc = self._lib.score_event_query(self.cuda_event)
        if rc not in (0, 1):
            check(rc, "score_event_query")
        return rc == 0

    def synchronize(self):
        check(self._lib.score_event_synchronize(self.cuda_event), "score_event_synchronize")

    def __del__(self):
        try:
            if self.cuda_event:
                self._lib.score_event_destroy(self.cuda_event)
        except Exception:
            pass


class ScoreHipError(RuntimeError):
    pass


def load():
    """Load (building first if needed) libscore_hip.so; raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SCORE_HIP_LIB") or LIB_PATH      # SCORE_HIP_LIB: another build of the same ABI (A/B runs)
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  Loaded before torch, this library would bind
    # the system's copy, and the process would hold two runtimes -- torch's streams and device state on one, these kernels'
    # launches on the other (seen as `score_context_create failed` when build() and smoke() shared a process)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
        fn.argtypes = args
        fn.restype = C.c_int64 if name.endswith("_bytes") or name.endswith("_floats") else C.c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kinds = {-1: "bad argument", -2: "unsupported shape", -3: "workspace too small",
                 -4: "a feature id outside [0, feature_size) was fed"}
        raise ScoreHipError("%s failed: %s" % (what, kinds.get(rc, "hipError_t %d" % rc)))


def make_config(feature_size, eb_dim, hidden_size, max_time_len, obj_per_time_slice, user_fnum, item_fnum,
                model_type="SCORE"):
    return Config(int(feature_size), int(eb_dim), int(hidden_size), int(max_time_len),
                  int(obj_per_time_slice), int(user_fnum), int(item_fnum), MODEL_TYPES[model_type])


def param_layout(cfg):
    """-> (entries [(name, offset, rows, cols, regularised, init)], n_floats, n_reg)"""
    lib = load()
    arr = (ParamEntry * 32)()
    nf, nr = C.c_int64(0), C.c_int64(0)
    n = lib.score_param_layout(C.byref(cfg), arr, 32, C.byref(nf), C.byref(nr))
    if n < 0:
        check(n, "score_param_layout")
    out = [(arr[i].name.decode(), arr[i].offset, arr[i].rows, arr[i].cols, arr[i].regularised, arr[i].init)
           for i in range(n)]
    return out, nf.value, nr.value


def workspace_layout(cfg, B):
    lib = load()
    ws = Workspace()
    check(lib.score_workspace_layout(C.byref(cfg), int(B), C.byref(ws)), "score_workspace_layout")
    return ws


def workspace_field(cfg, B, name):
    """(offset, offset of the second side / call or -1) of an internal workspace region, in floats"""
    lib = load()
    a, b = C.c_int64(0), C.c_int64(-1)
    check(lib.score_workspace_field(C.byref(cfg), int(B), name.encode(), C.byref(a), C.byref(b)), "score_workspace_field")
    return a.value, b.value


_listpack = None


def listpack():
    """The CPython extension that flattens nested feed lists into int32 (score_amd/cext/listpack.c), built in-tree
    with gcc on first use.  Host-side ingestion only -- no compute; returns None if it cannot be built (the caller
    then converts with NumPy, ~15x slower)."""
    global _listpack
    if _listpack is None:
        import importlib.machinery
        import importlib.util
        path = os.path.join(_HERE, "lib", "_listpack.so")
        try:
            if not os.path.exists(path):
                from . import build as _build
                _build.build_listpack()
            loader = importlib.machinery.ExtensionFileLoader("_listpack", path)
            spec = importlib.util.spec_from_file_location("_listpack", path, loader=loader)
            mod = importlib.util.module_from_spec(spec)
            loader.exec_module(mod)
            _listpack = mod
        except Exception:
            _listpack = False
    return _listpack or None
