"""Feasibility probe (timing only, results are NOT a valid training run): does the dense table sweep of step t hide
under step t+1's forward + backward when it runs as a small persistent grid on a low-priority stream?
  SCORE_ADAM_BLOCKS=<n> python tools/adam_overlap_probe.py [overlap|serial]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from score_amd.synth import make_world
from score_amd.model import SCORE
mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
w, kw = make_world("cfg3"); B = kw.pop("batch")
m = SCORE(seed=1, **kw)
m.table_flags.fill_(1)
bs = [m.device_batch(w.batch(B, i)) for i in range(8)]
main = torch.cuda.current_stream()
m.forward_backward(bs[0], 1e-4, 0.8)      # (the model's own side streams take their hardware queues first)
torch.cuda.synchronize()
if os.environ.get("PROBE_STREAM") == "plain":
    bg = torch.cuda.Stream(priority=0)
else:
    from score_amd.dist import _concurrent_stream
    bg = _concurrent_stream(torch.device("cuda:0"))      # a stream whose kernels really run beside the main stream's
ev4 = torch.cuda.Event(); ev4.record(main)
def step_tail(i):
    """the table sweep on the side stream from the moment the row gradients exist (stage boundary 4 of score_backward),
    i.e. beside the weight-gradient products that end the pass; exact (same inputs, same kernel)"""
    m.bwd_events = [None, None, None, None, ev4, None]
    m.forward_backward(bs[i % 8], 1e-4, 0.8)
    bg.wait_event(ev4)
    with torch.cuda.stream(bg):
        m.adam_table(1e-3)
        done = bg.record_event()
    m.adam_dense(1e-3, 1e-4)
    m.adam_advance()
    main.wait_event(done)
    return None
def step(i, prev_done):
    if mode == "tail":
        return step_tail(i)
    m.forward_backward(bs[i % 8], 1e-4, 0.8)
    m.adam_dense(1e-3, 1e-4)
    if mode == "serial":
        m.adam_table(1e-3)
        m.adam_advance()
        return None
    ev = main.record_event()
    if prev_done is not None:
        bg.wait_event(prev_done)
    bg.wait_event(ev)
    with torch.cuda.stream(bg):
        m._row_grads = True
        m.adam_table(1e-3)
        done = bg.record_event()
    m.adam_advance()
    return done
d = None
for i in range(10): d = step(i, d)
torch.cuda.synchronize()
t = time.perf_counter()
N = 200
for i in range(N): d = step(i, d)
torch.cuda.synchronize()
print(mode, os.environ.get("SCORE_ADAM_BLOCKS"), "%.3f ms/step" % ((time.perf_counter() - t) / N * 1e3))
