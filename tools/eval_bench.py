"""Evaluation pass at 1 + 99 candidates per target line (train_score.py:19, 144-163) on one MI355X: forward only
(model.eval_async), predictions / ids / labels kept on the device, ranking metrics + AUC / log-loss by
score_ranking_quality / score_auc_logloss.  Prints samples/s and the fused gather's live launch time; run it
under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` to see how much of the user side's 100-fold repetition
reaches HBM (user-side reuse)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
from score_amd import harness

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3")
ap.add_argument("--batch", type=int, default=2000)
ap.add_argument("--per-user", type=int, default=100)
ap.add_argument("--batches", type=int, default=4)
ap.add_argument("--passes", type=int, default=5)
a = ap.parse_args()
torch.cuda.set_device(0)
w, kw = make_world(a.config)
kw.pop("batch")
m = SCORE(seed=1111, **kw)
bs = [m.device_batch(w.batch(a.batch, 500 + i, per_user=a.per_user)) for i in range(a.batches)]
harness.evaluate_device(m, bs, 1e-4, a.per_user - 1)          # warm-up
torch.cuda.synchronize()
m.enable_stage_events(True)
t0 = time.perf_counter()
for _ in range(a.passes):
    res = harness.evaluate_device(m, bs, 1e-4, a.per_user - 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = a.passes * a.batches * a.batch
gather_ms = m.fwd_events[0].elapsed_time(m.fwd_events[1])
fwd_ms = m.fwd_events[0].elapsed_time(m.fwd_events[4])
print(json.dumps({"metric": "eval samples/sec (1+%d candidates per line, forward + device metrics)" % (a.per_user - 1),
                  "value": n / dt, "unit": "samples/s", "config": a.config, "batch": a.batch,
                  "ms_per_batch": dt / (a.passes * a.batches) * 1e3, "fwd_gather_coattn_ms": gather_ms,
                  "forward_ms": fwd_ms, "metrics": [round(float(x), 6) for x in res]}))
