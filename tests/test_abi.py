"""CPU tests of the C-ABI boundary: the library builds/loads and exports every symbol
include/score_hip.h declares; layouts agree with the oracle's parameter spec.
No compute calls (no GPU here)."""
import os
import re

import numpy as np
import pytest

from oracle import score_oracle as so
from score_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "score_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(score_\w+)\s*\(", src)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


@pytest.mark.parametrize("mt", so.MODEL_TYPES)
def test_param_layout_matches_oracle_spec(mt):
    # same variables, names, shapes, creation order and L2 filter as score.py (oracle param_spec)
    cfg = _lib.make_config(1000, 16, 32, 11, 10, 3, 4, mt)
    entries, n_w, n_reg = _lib.param_layout(cfg)
    spec = so.param_spec(so.Cfg(1000, 16, 32, 11, 10, 3, 4, mt))[1:]
    assert [e[0] for e in entries] == [s[0] for s in spec]
    for e, s in zip(entries, spec):
        shape = (e[2], e[3]) if e[3] else (e[2],)
        assert shape == tuple(s[1]), e[0]
        assert bool(e[4]) == s[3], e[0]
        assert {"zeros": 0, "ones": 1, "glorot": 2}[s[2]] == e[5]
        assert e[1] % 4 == 0
        assert (e[1] < n_reg) == bool(e[4])
    # no overlap
    spans = sorted((e[1], e[1] + e[2] * (e[3] or 1)) for e in entries)
    for a, b in zip(spans, spans[1:]):
        assert a[1] <= b[0]
    assert spans[-1][1] <= n_w


def test_workspace_layout_and_errors():
    cfg = _lib.make_config(1000, 16, 32, 11, 10, 3, 4, "SCORE")
    ws = _lib.workspace_layout(cfg, 200)
    assert ws.total_bytes > 0 and ws.loss > 0
    bad = _lib.make_config(1000, 15, 32, 11, 10, 3, 4, "SCORE")     # D not a multiple of 4
    with pytest.raises(_lib.ScoreHipError):
        _lib.workspace_layout(bad, 200)
    with pytest.raises(_lib.ScoreHipError):
        _lib.workspace_layout(cfg, 0)


def test_model_requires_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from score_amd.model import SCORE
    with pytest.raises(RuntimeError):
        SCORE(100, 4, 8, 3, 2, 3, 4)


def test_stripped_probe_kernels_cannot_reach_the_product_build():
    """the timing probes' wrong-by-design switches (no loads / no matrix instruction / no stores) stop the preprocessor
    unless -DSCORE_PROBE_BUILD is beside them, and build.py refuses that flag"""
    import subprocess
    from score_amd import build as b
    src = os.path.join(b.CSRC, "gru.hip")
    cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-E", "-DGRP_NOMFMA", src, "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode != 0 and "SCORE_PROBE_BUILD" in out.stderr
    saved = list(b.FLAGS)
    try:
        b.FLAGS.append("-DSCORE_PROBE_BUILD")
        with pytest.raises(RuntimeError):
            b.build()
    finally:
        b.FLAGS[:] = saved


def test_ctypes_structures_match_the_headers_layout():
    """every ctypes.Structure of score_amd/_lib.py against sizeof / late-field offsets of the C struct it stands for
    (score_abi_struct_sizes, host only): a field added on one side only shows here, not as a wrong pointer on the GPU"""
    import ctypes as C
    lib = _lib.load()
    out = (C.c_int64 * 32)()
    n = lib.score_abi_struct_sizes(out, 32)
    assert n == 14
    got = list(out)[:n]
    want = [None, C.sizeof(_lib.Config), C.sizeof(_lib.ParamEntry), C.sizeof(_lib.Batch), C.sizeof(_lib.Workspace), C.sizeof(_lib.Guard),
            C.sizeof(_lib.AdamTable), C.sizeof(_lib.State), C.sizeof(_lib.TrainStep), C.sizeof(_lib.Graph), C.sizeof(_lib.BatchOut),
            _lib.State.plan_workspace.offset, _lib.TrainStep.plan_stream.offset, _lib.AdamTable.skipped_steps.offset]
    names = ["score_step_scalars_t", "score_config_t", "score_param_entry_t", "score_batch_t", "score_workspace_t", "score_guard_t",
             "score_adam_table_t", "score_state_t", "score_train_step_t", "score_graph_t", "score_batch_out_t",
             "offsetof(score_state_t, plan_workspace)", "offsetof(score_train_step_t, plan_stream)",
             "offsetof(score_adam_table_t, skipped_steps)"]
    for k, (g, w) in enumerate(zip(got, want)):
        assert w is None or g == w, (names[k], g, w)
    assert lib.score_abi_struct_sizes(out, 3) == -1
