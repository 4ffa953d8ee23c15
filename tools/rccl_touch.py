"""First RCCL process of a box: one rank, one all_reduce and one all_to_all (what bench.py --force-sharded uses)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda"); y = torch.empty_like(x)
for n in (3, 50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        dist.all_reduce(x); dist.all_to_all_single(y, x)
    torch.cuda.synchronize(); print("%d rounds: %.1f us per pair" % (n, (time.perf_counter() - t) / n * 1e6), flush=True)
dist.destroy_process_group()
