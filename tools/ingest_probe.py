"""model.train fed the 8-tuple of NESTED PYTHON LISTS exactly as the reference's GraphLoader yields it
(graph_loader.py:383), per shape: samples/s end to end, and the host conversion's share (C walker vs np.asarray).
Run on the GPU box: python tools/ingest_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from score_amd.synth import make_world
from score_amd.model import SCORE, DeviceBatch
from score_amd import _lib
for cfg in ("tmall_default", "cfg2", "cfg3"):
    w, kw = make_world(cfg); B = kw.pop("batch")
    m = SCORE(seed=1, **kw)
    nested = [w.batch(B, i, as_lists=True) for i in range(3)]
    for b in nested: m.train(None, b, 1e-3, 1e-4)
    torch.cuda.synchronize()
    t = time.perf_counter(); n = 0
    for _ in range(3):
        for b in nested:
            m.train(None, b, 1e-3, 1e-4); n += 1
    torch.cuda.synchronize()
    s_lists = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for b in nested: DeviceBatch(m, b)
    torch.cuda.synchronize()
    s_conv = (time.perf_counter() - t) / len(nested)
    lp = _lib._listpack; _lib._listpack = False          # NumPy conversion for comparison
    t = time.perf_counter()
    for b in nested: DeviceBatch(m, b)
    torch.cuda.synchronize()
    s_np = (time.perf_counter() - t) / len(nested)
    _lib._listpack = lp
    dbs = [m.device_batch(b) for b in nested]
    for b in dbs: m.train_async(b, 1e-3, 1e-4)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        for b in dbs: m.train_async(b, 1e-3, 1e-4)
    torch.cuda.synchronize()
    s_dev = (time.perf_counter() - t) / 30
    print("%-14s B=%4d: nested lists %8.0f samples/s (%.2f ms/step; conversion+H2D %.2f ms with the C walker, %.2f ms with np.asarray); "
          "device-resident batches %8.0f samples/s (%.3f ms/step)" % (cfg, B, B / s_lists, s_lists * 1e3, s_conv * 1e3, s_np * 1e3, B / s_dev, s_dev * 1e3), flush=True)
