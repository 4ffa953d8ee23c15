"""Small-shape step time under bench.py's loop (forward_backward + apply_adam, no sync for N steps), with the batches
made two ways: through the host feed path (pinned staging, what bench.py does) or from device tensors.
  python tools/small_shape_host.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_amd.synth import make_world
from score_amd.model import SCORE
cfgname = sys.argv[1] if len(sys.argv) > 1 else "tmall_default"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w, kw = make_world(cfgname); B = kw.pop("batch")
for mode in ("host-feed", "device-tensors", "host-feed"):
    m = SCORE(seed=1, **kw)
    raw = [w.batch(B, i) for i in range(8)]
    if mode == "host-feed":
        bs = [m.device_batch(b) for b in raw]
    else:
        bs = [m.device_batch(tuple(torch.as_tensor(a).cuda() for a in b)) for b in raw]
    for i in range(10):
        m.forward_backward(bs[i % 8], 1e-4, 0.8); m.apply_adam(1e-3, 1e-4)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(N):
        m.forward_backward(bs[i % 8], 1e-4, 0.8); m.apply_adam(1e-3, 1e-4)
    host = time.perf_counter() - t
    torch.cuda.synchronize(); wall = time.perf_counter() - t
    print("%-15s host %.4f ms/step  wall %.4f ms/step  (threads alive: %d)" % (mode, host / N * 1e3, wall / N * 1e3, __import__("threading").active_count()), flush=True)
    del m, bs
