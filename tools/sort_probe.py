"""csrc/sort.hip alone (the index plan's sort for small batches): score_sort_pairs on cfg-3-shaped keys (2.87 M occurrences, 21-bit
row ids: 30 % dummy row, 20 % a dozen hot rows at the top of the id space, 30 % 15 k categorical rows, the rest uniform), against
torch.sort.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.   python tools/sort_probe.py [n] [key_bits]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from score_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_874_369
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 21
g = torch.Generator(device="cuda").manual_seed(1)
hi = 1 << bits
u = torch.rand((n,), device="cuda", generator=g)
keys = torch.randint(0, min(hi, 1_529_672), (n,), device="cuda", generator=g)
hot = min(hi, 1_529_672) - 1 - torch.randint(0, 12, (n,), device="cuda", generator=g)
cat = min(hi, 1_529_672) - 13 - torch.randint(0, 15000, (n,), device="cuda", generator=g)
keys = torch.where(u < 0.3, torch.zeros_like(keys), torch.where(u < 0.5, hot, torch.where(u < 0.8, cat, keys))).to(torch.int32)
vals = torch.arange(n, device="cuda", dtype=torch.int32)
k0, v0, k1, v1 = keys.clone(), vals.clone(), torch.empty_like(keys), torch.empty_like(vals)
tb = int(lib.score_sort_pairs_temp_bytes(n))
temp = torch.empty((tb,), dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
where = C.c_int32(0)
def run():
    k0.copy_(keys); v0.copy_(vals)
    assert lib.score_sort_pairs(P(k0), P(v0), P(k1), P(v1), n, bits, P(temp), tb, C.byref(where), st) == 0
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    k0.copy_(keys); v0.copy_(vals)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.score_sort_pairs(P(k0), P(v0), P(k1), P(v1), n, bits, P(temp), tb, C.byref(where), st)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("score_sort_pairs n=%d bits=%d: %.1f us (min %.1f)" % (n, bits, 1e3 * sorted(ts)[len(ts) // 2], 1e3 * min(ts)))
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.sort(keys, stable=True); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("torch.sort(stable) keys only: %.1f us" % (1e3 * min(ts)))
ko, vo = (k1, v1) if where.value == 1 else (k0, v0)
wk, order = torch.sort(keys.to(torch.int64), stable=True)
print("stable and sorted:", bool(torch.equal(ko.to(torch.int64), wk) and torch.equal(vo.to(torch.int64), order)))
