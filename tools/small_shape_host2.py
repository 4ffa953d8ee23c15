"""tools/small_shape_host.py plus bench.py's stage events on every 4th step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
w, kw = make_world("tmall_default"); B = kw.pop("batch")
N = 300
for mode in ("no-events", "events/4", "events/4+all-live", "no-events+all-live"):
    m = SCORE(seed=1111, **kw)
    bs = [m.device_batch(w.batch(B, i)) for i in range(8)]
    if "all-live" in mode:
        m.table_flags.fill_(1)
    for i in range(10):
        m.forward_backward(bs[i % 8], 1e-4, 0.8); m.apply_adam(1e-3, 1e-4)
    events = {}
    if not mode.startswith("no-events"):
        for i in range(0, N, 4):
            m.enable_stage_events(True)
            events[i] = (m.fwd_events, m.bwd_events) + tuple(torch.cuda.Event(enable_timing=True) for _ in range(4))
        m.enable_stage_events(False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(N):
        e0 = e1 = None
        if i in events:
            m.fwd_events, m.bwd_events, e0, e1 = events[i][:4]
            m.catchup_events = events[i][4:6]
        elif events:
            m.fwd_events = m.bwd_events = m.catchup_events = None
        m.forward_backward(bs[i % 8], 1e-4, 0.8)
        if e0 is not None: e0.record()
        m.apply_adam(1e-3, 1e-4)
        if e1 is not None: e1.record()
    host = time.perf_counter() - t
    torch.cuda.synchronize(); wall = time.perf_counter() - t
    print("%-20s host %.4f ms/step  wall %.4f ms/step" % (mode, host / N * 1e3, wall / N * 1e3), flush=True)
    del m, bs
