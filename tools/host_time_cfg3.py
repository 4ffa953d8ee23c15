"""How long does the host need to enqueue one cfg-3 step?  (bench.py small_shape_leg's loop on cfg-3 with few steps, so that the
queue never fills up and the enqueueing loop's time is the host's own.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
for steps in (10, 20, 40):
    print(json.dumps(bench.small_shape_leg(cfg, 1e-3, 1e-4, steps=steps, warmup=20)), flush=True)
