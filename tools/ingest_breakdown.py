"""Where the literal-interface step (model.train(sess, nested lists)) spends its time: conversion + upload vs the step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE, DeviceBatch
w, kw = make_world("cfg3"); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
nested = [w.batch(B, 70 + i, as_lists=True) for i in range(3)]
for i in range(3): m.train(None, nested[i], 1e-3, 1e-4)
import gc; gc.collect(); gc.freeze()
for nt in (8, 16):
    m.feed_threads = nt
    T = {"db_host": 0.0, "db_sync": 0.0, "step": 0.0, "whole": 0.0}
    N = 12
    for i in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); db = DeviceBatch(m, nested[i % 3]); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        m.train(None, db, 1e-3, 1e-4); t3 = time.perf_counter()
        T["db_host"] += t1 - t0; T["db_sync"] += t2 - t1; T["step"] += t3 - t2
    for i in range(N):
        t0 = time.perf_counter(); m.train(None, nested[i % 3], 1e-3, 1e-4); T["whole"] += time.perf_counter() - t0
    print("threads %d:" % nt, {k: round(v / N * 1e3, 3) for k, v in T.items()}, "ms")
