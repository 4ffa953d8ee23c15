"""Where a sharded step's time goes (one rank through RCCL): GPU events on the main stream at the phase
boundaries + host timestamps, with and without the next-batch plan prefetch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from score_amd.synth import make_world
from score_amd.dist import ShardedSCORE
w, kw = make_world("cfg3"); B = kw.pop("batch")
m = ShardedSCORE(seed=1, **kw)
NB = int(os.environ.get('TL_BATCHES', '3'))
bs = [m.device_batch(w.batch(B, i)) for i in range(NB)]
be, cm = m.backend, m.comm
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for prefetch in ((True,) if os.environ.get("TL_ONLY") else (False, True)):
    for i in range(4): m.train_async(bs[i % NB], 1e-3, 1e-4, next_batch=bs[(i + 1) % NB] if prefetch else None)
    if os.environ.get('TL_BARRIER'): dist.barrier()
    torch.cuda.synchronize()
    N = int(os.environ.get('TL_STEPS', '12')); recs = []
    t_all = time.perf_counter()
    for i in range(N):
        batch, nxt = bs[i % NB], (bs[(i + 1) % NB] if prefetch else None)
        if os.environ.get("TL_EVENTS"): m.enable_stage_events(True)
        if os.environ.get("TL_FB"):
            h0 = time.perf_counter(); e0 = ev()
            m.forward_backward(batch, 1e-4, 0.8, None, nxt)
            e4 = ev(); m.apply_adam(1e-3, 1e-4); h4 = time.perf_counter(); e5 = ev()
            recs.append((e0, e0, e0, e0, e4, e5, h0, h0, h0, h0, h4))
            continue
        h0 = time.perf_counter(); e0 = ev()
        plan, mini = m._fetch(batch)
        h1 = time.perf_counter(); e1 = ev()
        if nxt is not None: m._prefetch_launch(nxt)
        be.set_global_batch(B)
        fw = be.forward(plan, mini, 1e-4, 0.8, None)
        e2 = ev()
        mini_g = be.backward(plan, mini, fw, 0.8)
        h2 = time.perf_counter(); e3 = ev()
        if nxt is not None: m._prefetch_finish()
        h3 = time.perf_counter()
        grads_in = torch.empty((plan["req"].numel(), m.D), dtype=torch.float32, device=m.device)
        cm.all_to_all(grads_in, mini_g, plan["recv"], plan["send"])
        cm.all_reduce_sum(be.dense_grad())
        be.accumulate(plan["req"], grads_in, plan["recv"])
        loss = fw["loss"].clone(); cm.all_reduce_sum(loss[1:2])
        e4 = ev()
        m.apply_adam(1e-3, 1e-4)
        h4 = time.perf_counter(); e5 = ev()
        recs.append((e0, e1, e2, e3, e4, e5, h0, h1, h2, h3, h4))
    torch.cuda.synchronize(); wall = (time.perf_counter() - t_all) / N * 1e3
    g = lambda a, b: sum(r[a].elapsed_time(r[b]) for r in recs[2:]) / (N - 2)
    hh = lambda a, b: sum(r[b] - r[a] for r in recs[2:]) / (N - 2) * 1e3
    nxt_gap = sum(recs[i][5].elapsed_time(recs[i + 1][0]) for i in range(2, N - 1)) / (N - 3)
    print("prefetch=%s wall %.3f ms/step | GPU: fetch %.3f fwd %.3f bwd %.3f tail %.3f adam %.3f gap-to-next %.3f | "
          "host: fetch %.3f enqueue-fwd-bwd %.3f prefetch-finish %.3f tail+adam %.3f"
          % (prefetch, wall, g(0, 1), g(1, 2), g(2, 3), g(3, 4), g(4, 5), nxt_gap, hh(6, 7), hh(7, 8), hh(8, 9), hh(9, 10)))
dist.destroy_process_group()
