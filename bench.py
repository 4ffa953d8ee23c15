#!/usr/bin/env python3
"""bench.py -- train samples/sec of the SCoRe hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or called
   plainly -- `python bench.py --gpus N` then starts that launcher itself as a child process, before touching the GPU)

A "step" is one full training step of SCORE (score.py:101-116: forward, backward, dense TF-Adam over the
whole table and all dense variables) on one synthetic Tmall-shaped batch whose int32 index tensors are
already resident in HBM.  Prints ONE JSON line (rank 0).

`value` is the STEADY STATE of a long run: every table row carries Adam moments (dense ApplyAdam then owes every
row an update every step -- in a short run from a fresh optimizer only the rows touched so far).  The table
optimizer is the time-tiled one (score_amd/csrc/adam_tiled.hip: bit-identical to the per-step sweep, which
`value_dense_adam_sweep` times in the same process; updates still owed after the last step are applied inside
the timed region), all other state as the loader produces it (length = T - 2 for every sample, mirroring the reference's
train split 9 of 11, graph_loader.py:382; the slices every sample masks are skipped).  Beside it, in the same
line and measured in the same process: `value_best_case` (fresh optimizer state), `value_all_slices` (nothing
skipped), `ingestion` (device-side batch assembly inside the loop; nested Python lists as the reference feeds
them), `roofline` (the fused embedding-gather + co-attention forward kernel -- the dominant HBM-bound kernel -- on a
LOW-DUPLICATION batch over a table far larger than the Infinity Cache, where the algorithmic bytes of SURVEY.md 8(d)
are what memory has to deliver: frac <= 1 and reproducible from profiles/r03_cfg3_gather_probe_kernel_stats.csv;
HIP-event duration on the launch stream, PMC traffic of the committed profile of the same build),
`roofline_bench_workload` (the same kernel on the loader-shaped bench batches, whose repeats the caches serve),
`roofline_other` (row scatter and table optimizer, with PMC traffic), `cpu_baseline` (CPU restatement of the TF graph,
"port") and `cpu_baseline_literal_tile` (the materialised [B,T,K,K,3D] form TF really executes, at the Tmall-default
shape).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak (MI355X_MICROARCH.md; AMD's headline doubles it with sparsity)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable (float4 copy)
PROFILE_ROUND = os.environ.get("SCORE_PROFILE_ROUND", "r06")


def alg_bytes_per_sample(T, K, D, Fu, Fi):
    """fused gather forward = idx + row reads + reduced outputs."""
    R = 2 * T * K * (Fu + Fi) + (Fu + Fi)
    Du, Di = Fu * D, Fi * D
    out = 4 * (T * 2 * (Du + Di) + (Du + Di) + T * 4 * K)
    return R * 4 + R * 4 * D + out, R


def usable_cpus():
    """Host cores this process may actually use: affinity mask and cgroup quota, not os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(kw, batch, params, budget_s=25.0, max_steps=4, max_threads=32, tiled=False):
    """CPU restatement of the TF1 graph (oracle/: collapsed co-attention, or tiled=True the literal materialised
    tile) + dense TF-Adam, full train step, on a bounded sample: the first step is timed too and is the whole
    sample if it alone exceeds the budget."""
    from oracle import score_oracle as so
    threads = min(usable_cpus(), max_threads)
    torch.set_num_threads(threads)
    m = so.OracleModel(kw["feature_size"], kw["eb_dim"], kw["hidden_size"], kw["max_time_len"],
                       kw["obj_per_time_slice"], kw["user_fnum"], kw["item_fnum"], "SCORE", params=params, tiled=tiled)
    B = len(batch[6])
    t0 = time.time()
    m.train(None, batch, 1e-3, 1e-4, keep_prob=1.0)
    first = time.time() - t0
    times = []
    t_all = time.time()
    while first < budget_s and len(times) < max_steps and (time.time() - t_all) < budget_s:
        t0 = time.time()
        m.train(None, batch, 1e-3, 1e-4, keep_prob=1.0)
        times.append(time.time() - t0)
    med = float(np.median(times)) if times else first
    form = ("literal materialised [B,T,K,K,3D] co-attention tile (score.py:147-167 as TF executes it)" if tiled
            else "collapsed co-attention")
    return {"value": B / med, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d full train step(s) (fwd+bwd+dense TF-Adam, %s) after one %s step, B=%d, N=%d rows, T=%d, K=%d, "
                      "D=%d, H=%d, median %.3f s/step; CPU restatement of the TF1 graph (TensorFlow unavailable), "
                      "torch-CPU fp32, %d threads (os.cpu_count()=%d, usable=%d)"
                      % (max(len(times), 1), form, "untimed warm-up" if times else "(timed, no warm-up)", B,
                         kw["feature_size"], kw["max_time_len"], kw["obj_per_time_slice"], kw["eb_dim"],
                         kw["hidden_size"], med, threads, os.cpu_count(), usable_cpus())}


def src_sha(name):
    with open(os.path.join(ROOT, "score_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def committed_traffic(config, key, kernel_substr, sources=("embed.hip",)):
    """HBM bytes per launch of a kernel (kernel_substr a string) or per step of a group of kernels (a tuple of
    substrings: every kernel whose name contains one of them, bytes x launches per step summed) from a committed PMC
    profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 FETCH_SIZE x2 correction:
    tools/summarize_profile.py) -- only if that profile was taken on THIS build of the source files the kernels live in
    (`sources`); otherwise None (a stale profile is dropped, not quoted)."""
    path = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (PROFILE_ROUND, config))
    try:
        pj = json.load(open(path))
        sec = pj[key]
        shas = sec.get("source_sha16") or {"embed.hip": sec.get("embed_hip_sha16")}
        for src in sources:
            if shas.get(src) != src_sha(src):
                return None, {"dropped": "profile %s was taken on another build of %s" % (os.path.basename(path), src)}
        info = {"profile": os.path.relpath(path, ROOT), "commit": sec.get("commit"), "workload": sec.get("workload"),
                "source_sha16": {s_: shas.get(s_) for s_ in sources}}
        if isinstance(kernel_substr, str):
            for k, v in sec["kernels"].items():
                if kernel_substr in k:
                    return v["hbm_bytes"], info
            return None, None
        steps = float(sec.get("steps_profiled") or 0)
        if steps <= 0:
            return None, None
        tot, used = 0.0, {}
        for k, v in sec["kernels"].items():
            if any(sub in k for sub in kernel_substr):
                per_step = max(1, int(round(v["dispatches"] / steps)))     # (the profiled run holds one extra backward pass)
                tot += v["hbm_bytes"] * per_step
                used[k.split("<")[0]] = used.get(k.split("<")[0], 0) + v["hbm_bytes"] * per_step
        info["kernels_bytes_per_step"] = used
        return (tot if used else None), (info if used else None)
    except Exception:
        pass
    return None, None


def event_pair_overhead_ms():
    """what an EMPTY pair of timing events measures on this stream: stage durations include it, rocprofv3's do not"""
    cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    pad = torch.zeros((1 << 20,), device="cuda")
    for a_, b_ in cal:
        pad.add_(1.0)
        a_.record()
        b_.record()
    torch.cuda.synchronize()
    return float(np.median([a_.elapsed_time(b_) for a_, b_ in cal]))


def gather_probe(model, kw, B, n_probe_rows, iters, seed=0):
    """The fused gather + co-attention forward kernel on a LOW-DUPLICATION batch (score_amd.synth.lowdup_batch):
    uniform ids over a probe table of n_probe_rows rows (far beyond the 256 MiB Infinity Cache), no dummy slices,
    nothing shared.  Runs score_forward (eval) and reads the gather stage from the HIP events on the launch stream."""
    import ctypes as C
    from score_amd import _lib
    from score_amd.synth import lowdup_batch
    T, K, D, Fu, Fi = kw["max_time_len"], kw["obj_per_time_slice"], kw["eb_dim"], kw["user_fnum"], kw["item_fnum"]
    table = torch.empty((n_probe_rows, D), dtype=torch.float32, device=model.device)
    _lib.check(model.lib.score_table_init(C.c_void_p(table.data_ptr()), n_probe_rows, D, 1, 0, n_probe_rows,
                                          C.c_uint64(12345), model._stream()), "score_table_init")
    dbs = [model.device_batch(lowdup_batch(n_probe_rows, B, T, K, Fu, Fi, seed + i)) for i in range(4)]
    saved, saved_rows = model.table, None
    times = []
    try:
        model.table = table                      # score_state_t.table -> the probe table (forward only: no optimizer)
        for i in range(iters + 3):
            model.enable_stage_events(True)
            ev = model.fwd_events
            model._forward(dbs[i % 4], 1e-4, 1.0, None)
            torch.cuda.synchronize()
            if i >= 3:
                times.append(ev[0].elapsed_time(ev[1]))
    finally:
        model.table = saved
        model.enable_stage_events(False)
    del table
    torch.cuda.empty_cache()
    ab, R = alg_bytes_per_sample(T, K, D, Fu, Fi)
    # distinct rows a batch of uniform draws names, in expectation: what must come from memory at least once
    uses = R * B
    uniq = n_probe_rows * (1.0 - np.exp(-uses / float(n_probe_rows)))
    return {"ms": float(np.mean(times)), "ms_min": float(np.min(times)), "alg_bytes": ab * B, "row_uses": uses,
            "expected_distinct_rows": int(uniq), "compulsory_row_bytes": int(uniq) * 4 * D, "probe_rows": n_probe_rows}


def panel_gemm_probe(model, kw, B, A, iters=20):
    """The GRU input projections of both sides as the step runs them (csrc/gemm_panel.hip through its C-ABI op,
    score_gemm_panel_run: weights as prepared fragment images) on synthetic operands of the workload's shape:
    2 x [B*A, I] . [I, 3H] + bias, A = the active time slices.  Duration by HIP events on the launch stream."""
    import ctypes as C
    from score_amd import _lib
    D, H, Fu, Fi = kw["eb_dim"], kw["hidden_size"], kw["user_fnum"], kw["item_fnum"]
    M, N, K = B * A, 3 * H, D * (Fu + Fi)
    dev = model.device
    a = [torch.randn((M, K), device=dev) for _ in range(2)]
    w = [torch.randn((K, N), device=dev) * 0.05 for _ in range(2)]
    bias = [torch.randn((N,), device=dev) for _ in range(2)]
    c = [torch.empty((M, N), device=dev) for _ in range(2)]
    images = torch.empty((2 * (K // 32) * 8 * ((N + 127) // 128) * 768,), device=dev)
    arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = model.lib.score_gemm_panel_images(0, 2, N, K, arr(w), N, C.c_void_p(images.data_ptr()), images.numel(), st)
    if rc != 0:
        return None                         # (shape not covered: the tiled kernels run it)
    Aa, Ca, Ba = arr(a), arr(c), arr(bias)
    run = lambda: model.lib.score_gemm_panel_run(2, M, N, K, Aa, K, Ca, N, Ba, C.c_void_p(images.data_ptr()), images.numel(), st)
    if run() != 0:
        return None
    for _ in range(iters):                  # (warm: the legs before this one end in host work; the clock has dropped)
        run()
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        best, tot = min(best, t), tot + t
    ref = a[0][:256].double() @ w[0].double() + bias[0].double()
    err = float((c[0][:256].double() - ref).abs().max() / ref.abs().max())
    return {"ms": tot / 3, "ms_min": best, "M": M, "N": N, "K": K, "max_rel_err_vs_fp64": err,
            "flops": 2.0 * 2 * M * N * K, "a_bytes": 2 * M * K * 4, "c_bytes": 2 * M * N * 4, "w_bytes": 2 * K * N * 4}


def stream_copy_ceiling(model, n_bytes=1 << 30, iters=10):
    """SURVEY.md 8(d)'s second roofline denominator, measured on THIS box in THIS run: a plain float4 stream copy
    (score_stream_copy, include/score_hip.h) between two buffers of n_bytes each -- far beyond the 256 MiB Infinity Cache --,
    timed with HIP events on the launch stream.  GB/s of read + write traffic."""
    import ctypes as C
    from score_amd import _lib
    n = n_bytes // 4
    src = torch.empty((n,), dtype=torch.float32, device=model.device).normal_()
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _lib.check(model.lib.score_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, st),
                             "score_stream_copy")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / iters)
    ok = bool(torch.equal(dst[:4096], src[:4096]) and torch.equal(dst[-4096:], src[-4096:]))
    del src, dst
    torch.cuda.empty_cache()
    ms = float(np.median(times))
    return {"GBs": 2.0 * n_bytes / (ms * 1e-3) / 1e9, "ms": ms, "bytes_each_way": n_bytes, "copy_verified": ok,
            "what": "float4 grid-stride copy kernel (score_stream_copy), %d MiB read + %d MiB written per launch, median of "
                    "3 x %d launches by HIP events" % (n_bytes >> 20, n_bytes >> 20, iters)}


class Heartbeat(object):
    """Several ranks: a hung collective must not eat the driver's whole timeout in silence.  A daemon thread prints one
    line per rank to stderr every `every` seconds (phase, step, the communicator's last collective) and, when the main
    thread has made no progress for `stall` seconds, says which rank hangs where and ends the process with exit code 3 --
    the launcher (torch.distributed.run) then takes the other ranks down.  (The process exits; nothing is re-executed.)"""

    def __init__(self, rank, comm_of, every=20.0, stall=150.0):
        import threading
        self.rank, self.comm_of, self.every, self.stall = rank, comm_of, every, stall
        self.phase, self.count, self._seen, self._t_seen = "start", 0, None, time.time()
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, name="bench-heartbeat", daemon=True)
        self._t.start()

    def tick(self, phase=None):
        if phase is not None:
            self.phase = phase
        self.count += 1

    def stop(self):
        self._stop.set()

    def _run(self):
        last_print = time.time()
        while not self._stop.wait(1.0):
            now = time.time()
            state = (self.phase, self.count)
            if state != self._seen:
                self._seen, self._t_seen = state, now
            cm = self.comm_of()
            coll = getattr(cm, "last", None) if cm is not None else None
            if now - last_print >= self.every:
                last_print = now
                sys.stderr.write("bench.py heartbeat: rank %d phase=%s ticks=%d last_collective=%s idle=%.0fs\n"
                                 % (self.rank, self.phase, self.count, coll, now - self._t_seen))
                sys.stderr.flush()
            # (set-up and reporting are single long host-side stretches -- synthetic batches, table initialisation, on a fresh
            #  box the first import of torch: three times the patience there)
            limit = self.stall * (3.0 if self.phase in ("start", "setup", "report", "after") else 1.0)
            if now - self._t_seen > limit:
                sys.stderr.write("bench.py: rank %d made no progress for %.0f s in phase '%s' (tick %d), last collective "
                                 "entered: %s -- giving up (exit 3)\n" % (self.rank, now - self._t_seen, self.phase,
                                                                           self.count, coll))
                sys.stderr.flush()
                os._exit(3)


def small_shape_leg(config, lr, reg_lambda, steps=1000, warmup=100, n_batches=8):
    """One short leg of `model.train_async` on another shape of the path (the reference's own B = 200 / D = 16 / H = 32 shape,
    train_score.py:15-16,371-372, and BASELINE.json configs[1]): a model of its own, every table row live, device-resident
    batches with the next one announced (as the headline does), `warmup` untimed steps, `steps` timed between two
    synchronisations with the optimizer's flush inside.  host_us_per_step: this thread's time in the enqueueing loop alone (the
    step is bound by it at these shapes); p50: one HIP event per step over a stretch right behind the timed one."""
    from score_amd.synth import make_world
    from score_amd.model import SCORE
    world, kw = make_world(config)
    B = kw.pop("batch")
    model = SCORE(seed=1111, **kw)
    model.table_flags.fill_(1)
    batches = [model.device_batch(world.batch(B, i)) for i in range(n_batches)]
    # (as the headline does before its warm-up: what the process has alive by now -- the headline's model is gone, its garbage is
    #  not -- goes to the permanent generation, so that no generation-2 pass of the collector lands in a host-bound timed loop)
    import gc
    gc.collect()
    gc.freeze()

    def run(n, marks=None):
        for i in range(n):
            if marks is not None:
                marks[i].record()
            nb = batches[(i + 1) % n_batches] if i + 1 < n else None
            model.train_async(batches[i % n_batches], lr, reg_lambda, 0.8, None, nb)
        if marks is not None:
            marks[n].record()
    run(warmup)
    model._flush_adam()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    t_host = time.perf_counter() - t0
    model._flush_adam()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_p = min(steps, 100)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_p + 1)]
    run(n_p, marks)
    torch.cuda.synchronize()
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_p)]
    launches = None
    try:            # the launch count of one step of this shape: first line of the committed kernel sequence of this round
        with open(os.path.join(ROOT, "profiles", "%s_%s_sequence.txt" % (PROFILE_ROUND, config))) as f:
            launches = int(f.readline().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    out = {"samples_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 5),
           "ms_p50": round(float(np.median(per)), 5), "host_us_per_step": round(t_host / steps * 1e6, 1),
           "launches_per_step": launches, "B": B, "steps": steps, "warmup": warmup,
           "form": "per-sample" if model.persample_form(B, batches[0].active_slices) else "layered"}
    del model, batches
    torch.cuda.empty_cache()
    return out


def visible_gpus():
    """GPUs this process could use, counted WITHOUT touching the HIP runtime: the KFD topology in sysfs (a node with
    simd_count > 0 is a GPU), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  0 without a KFD driver; None if the topology cannot be read."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(base):
        return 0                      # no KFD driver: no ROCm device can be opened
    try:
        n = readable = 0
        for d in os.listdir(base):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, d, "properties")) if len(line.split()) >= 2)
            except Exception:
                continue
            readable += 1
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if readable == 0:
            return None               # (nodes exist but none can be read: unknown -- the ranks themselves refuse a missing device)
    except Exception:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` as a child process (one rank per GPU
    over RCCL; rank 0 prints the one JSON line to the inherited stdout) and return its exit code.  This process never
    touches the GPU: the devices are counted from the KFD topology in sysfs (visible_gpus; when that is unavailable the
    count is left to the ranks themselves, which refuse a missing device)."""
    import socket
    import subprocess
    rehearsal = os.environ.get("SCORE_BENCH_DEVICE") is not None      # several ranks on ONE device (gloo rehearsal)
    n_dev = visible_gpus()
    if n_dev is not None and n_dev < n and not rehearsal:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this box: not started (nothing has touched the "
                         "GPU).  To rehearse N ranks on one device: SCORE_BENCH_DEVICE=0 SCORE_DIST_BACKEND=gloo "
                         "python bench.py --gpus %d\n" % (n, n_dev, n))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batches", type=int, default=8, help="distinct pre-staged batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements (best case, all slices, gather probe, ingestion): headline only")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded all-to-all path even with one rank (exercises RCCL plumbing)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="sharded path: do NOT start the next batch's index plan under this step's compute")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="sharded path: optimizer after the backward pass instead of the pipelined step (table update "
                         "and the next batch's row fetch under this step's weight-gradient tail and dense all-reduce)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: the global batch is fixed at this many samples and every rank trains "
                         "global_batch / N of them (SURVEY 8d cfg-4: global B = 1024).  Default: weak scaling, every "
                         "rank trains the config's batch")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the HIP stage events (live kernel timing for `roofline`) on every E-th timed step: "
                         "eleven timing events per step cost ~3 %% of a 1.8 ms step")
    ap.add_argument("--step-marks-timed", action="store_true",
                    help="one event mark per step INSIDE the timed region (default: only its first and last step; ms_per_step_p50 "
                         "from a separate stretch of steps behind it)")
    ap.add_argument("--stage-events-timed", action="store_true",
                    help="record ALL stage boundary events inside the timed region (every E-th step) instead of on extra steps "
                         "behind it: the round-1..3 protocol (costs ~0.07 ms on each step that carries them)")
    ap.add_argument("--fresh-state", action="store_true",
                    help="headline from a FRESH optimizer state (only rows touched during the run carry moments): the "
                         "best case; default is the steady state with every table row live")
    ap.add_argument("--no-skip-masked", action="store_true",
                    help="gather and compute all T time slices, also those past every sample's length (whose "
                         "results the model masks): A/B for score_batch_t.active_slices")
    ap.add_argument("--no-look-ahead", dest="look_ahead", action="store_false",
                    help="do NOT tell apply_adam which batch comes next (SCOREBASE.apply_adam(next_batch=): with the time-tiled "
                         "table optimizer the next batch's rows are brought up to date beside this step's weight-gradient "
                         "products instead of in front of the next forward pass)")
    ap.add_argument("--set", dest="model_attrs", action="append", default=[], metavar="ATTR=VALUE",
                    help="set a public attribute of the model before the run (A/B of its placement choices: loss_on_side=0, "
                         "adam_sweep_at=3, overlap_finishers_min_rows=0 ...); the value is parsed as a Python literal")
    ap.add_argument("--debug-flags", type=int, default=0, help="score_state_t.debug_flags (A/B switches of the launch sequence)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured hipGraph (SCOREBASE.enable_graph: launch-bound small shapes); "
                         "the timed loop then carries no stage events -- stages_ms come from a few eager steps after it")
    ap.add_argument("--gather-probe-only", action="store_true",
                    help="run only the low-duplication gather probe (for rocprofv3 --pmc passes) and print its JSON")
    ap.add_argument("--probe-rows", type=int, default=32_000_000)
    ap.add_argument("--a2a", choices=["split", "remote", "probe"], default=None,
                    help="form of the row / gradient all-to-all with N > 1 ranks: 'remote' = list form with empty own slots (a rank's own "
                         "rows never go through RCCL), 'split' = all_to_all_single with the own segment inside (the default: the plain RCCL "
                         "path), 'probe' = try both at set-up and take the list form if it round-trips on every rank "
                         "(score_amd/dist.py TorchDistComm.probe_a2a)")
    ap.add_argument("--small-shape-leg", default=None, metavar="CONFIG",
                    help="(internal) run ONE small-shape leg on CONFIG in this process and print its JSON: the default run starts "
                         "one child process per leg")
    ap.add_argument("--no-small-shapes", action="store_true",
                    help="skip the two short legs on cfg-2 and the reference's Tmall-default shape (`small_shapes`, the LAST key of the line)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--reg-lambda", type=float, default=1e-4)
    args = ap.parse_args()

    if args.small_shape_leg:
        torch.cuda.set_device(int(os.environ.get("SCORE_BENCH_DEVICE", "0")))
        print(json.dumps(small_shape_leg(args.small_shape_leg, args.lr, args.reg_lambda)), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called as plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing above this line has touched
        # the GPU (no torch.cuda call that initialises HIP, no _lib.load()): the ranks are CHILD processes of
        # torch.distributed.run, this process only waits for them and hands their exit code on
        sys.exit(spawn_ranks(args.gpus))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or without "
                         "torch.distributed.run: `python bench.py --gpus N` starts its own ranks)"
                         % (args.gpus, world_size, args.gpus))
    # rehearsal aids (several ranks on a one-GPU box): SCORE_BENCH_DEVICE pins every rank to one device,
    # SCORE_DIST_BACKEND=gloo swaps RCCL for gloo (device tensors staged through host memory, score_amd/dist.py)
    if os.environ.get("SCORE_BENCH_DEVICE"):
        local_rank = int(os.environ["SCORE_BENCH_DEVICE"])
    backend = os.environ.get("SCORE_DIST_BACKEND", "nccl")
    if world_size > 1 or args.force_sharded:
        # The sharded step runs five streams (main, the engine's side stream, gradient exchange, index prefetch,
        # RCCL's own).  HIP deals streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues in first-use order; two
        # of them on one queue serialise, and which two changes from process to process: 1.89 - 2.30 ms/step with 4
        # queues, 1.77 every time with 8 (profiles/r02_probes.md).  Read when the HIP runtime starts: set before it.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world_size > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that never completes must fail within minutes, not at the default 10 (= the driver's whole limit)
        from datetime import timedelta
        tmo = timedelta(seconds=int(os.environ.get("SCORE_DIST_TIMEOUT_S", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, timeout=tmo,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size, timeout=tmo)

    from score_amd.synth import make_world
    from score_amd.model import SCORE
    hb = None
    model = None
    if dist is not None and world_size > 1:
        hb = Heartbeat(rank, lambda: getattr(model, "comm", None))
    beat = (lambda phase=None: hb.tick(phase)) if hb is not None else (lambda phase=None: None)
    beat("setup")
    world, kw = make_world(args.config)
    B = kw.pop("batch")
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % (2 * world_size):
            raise SystemExit("--global-batch must be a multiple of 2 * N (whole target lines per rank)")
        B = args.global_batch // world_size
    T, K, D = kw["max_time_len"], kw["obj_per_time_slice"], kw["eb_dim"]
    Fu, Fi = kw["user_fnum"], kw["item_fnum"]

    if sharded:
        from score_amd.dist import ShardedSCORE
        if args.a2a:
            os.environ["SCORE_A2A"] = args.a2a          # (read by the set-up probe of every rank)
        model = ShardedSCORE(seed=1111, **kw)
    else:
        model = SCORE(seed=1111, **kw)
    inner = model.backend.m if sharded else model     # owns the table (shard) and its optimizer state

    if args.gather_probe_only:
        pr = gather_probe(inner, kw, B, args.probe_rows, 20)
        pr["achieved_GBs_algorithmic"] = pr["alg_bytes"] / (pr["ms"] * 1e-3) / 1e9
        print(json.dumps(pr), flush=True)
        return

    if args.no_skip_masked:
        inner.skip_masked_slices = False
    if args.debug_flags:
        inner.debug_flags = args.debug_flags
    for kv in args.model_attrs:
        import ast
        k, _, v = kv.partition("=")
        if k.startswith("_") or not hasattr(inner, k):
            raise SystemExit("bench.py --set: the model has no public attribute %r" % k)
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass                # a bare word: kept as a string
        setattr(inner, k, v)
    # every rank trains on its own batches (weak: B each, global B * N; strong: global_batch / N each)
    batches = [model.device_batch(world.batch(B, rank * 1000 + i)) for i in range(args.batches)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    graph = args.graph and not sharded
    look_ahead = args.look_ahead and not sharded and not graph
    if graph:
        model.enable_graph(True)
    last_loss = [None]

    mark_every_step = True

    def run_steps(n, first=0, events=None, step_marks=None):
        """step_marks: a list of n + 1 timing events -- [i] is recorded on the launch stream before step i, [n] after the last"""
        fb = None
        if graph and events is None:
            for i in range(first, first + n):
                if step_marks is not None:
                    step_marks[i - first].record()
                last_loss[0] = model.train_async(batches[i % len(batches)], args.lr, args.reg_lambda)
            if step_marks is not None:
                step_marks[n].record()
            return None
        for i in range(first, first + n):
            beat()
            if step_marks is not None and (mark_every_step or i == first):
                step_marks[i - first].record()
            e_a0 = e_a1 = None
            if events is not None and i in events:
                model.fwd_events, model.bwd_events, e_a0, e_a1 = events[i][:4]
                inner.catchup_events = events[i][4:6] if events[i][4] is not None else None
            elif events is not None:
                model.fwd_events = model.bwd_events = inner.catchup_events = None
            if sharded:   # optionally run the next batch's index-only phase (plan + row requests) inside this step
                nxt = batches[(i + 1) % len(batches)] if (args.prefetch and i + 1 < first + n) else None
                # with a next batch the step is pipelined: the optimizer runs inside (apply_adam below is then a no-op)
                fb = model.forward_backward(batches[i % len(batches)], args.reg_lambda, 0.8, None, nxt,
                                            lr=args.lr if args.pipeline else None)
            elif events is None or i not in events or (events[i][1] is None and events[i][2] is None and events[i][4] is None):
                # (also the steps of the timed region that carry the dominant kernel's two timing events: forward stage events only)
                # the model's own one-call step (score.py:101-116 is one sess.run): forward_backward + apply_adam inside -- and, in the
                # steady state of the per-sample form, one call into the library for the whole step (score_train_step)
                # (no batch is announced behind the last step of a run: none comes, and the look-ahead work queued for it
                #  would only be waited for by the optimizer's flush that closes the timed region)
                nb = batches[(i + 1) % len(batches)] if (look_ahead and i + 1 < first + n) else None
                last_loss[0] = model.train_async(batches[i % len(batches)], args.lr, args.reg_lambda, 0.8, None, nb)
                fb = inner._workspace(batches[i % len(batches)].B) if hasattr(batches[i % len(batches)], "B") else fb
                continue
            else:
                fb = model.forward_backward(batches[i % len(batches)], args.reg_lambda, 0.8)
            if e_a0 is not None:
                e_a0.record()
            if look_ahead and i + 1 < first + n:      # the next batch is known (the loader's queue): its rows are caught up beside this step's tail
                model.apply_adam(args.lr, args.reg_lambda, next_batch=batches[(i + 1) % len(batches)])
            else:
                model.apply_adam(args.lr, args.reg_lambda)
            if e_a1 is not None:
                e_a1.record()
        if step_marks is not None:
            step_marks[n].record()
        return fb

    def finish_adam():
        """time-tiled table optimizer: apply every update still owed (score_adam_catchup_rows over the whole table)"""
        f = getattr(inner, "_flush_adam", None)
        if f is not None:
            f()
    tiled = bool(getattr(inner, "_tiled_on", lambda: False)()) and not graph

    # ---------------------------------------------------------------- headline: steady state
    if not args.fresh_state:
        inner.table_flags.fill_(1)      # every row carries Adam moments: the state a long run converges to
    events = {}
    every = max(1, args.event_every)

    def full_event_set():
        model.enable_stage_events(True)
        return (model.fwd_events, model.bwd_events) + tuple(torch.cuda.Event(enable_timing=True) for _ in range(4))
    # Inside the timed region only the dominant kernel is bracketed (the fused gather: two events on every E-th step,
    # `roofline_bench_workload`'s live launch duration); the full stage table -- eleven events per step, +0.07 ms on a step that
    # carries them (round 4: per-step marks, 1.35 vs 1.27 ms) -- comes from a few extra steps right behind the timed region
    for i in range(args.steps):
        if i % every:
            continue
        if args.stage_events_timed:
            events[i] = full_event_set()
        else:
            model.enable_stage_events(True)
            events[i] = (model.fwd_events[:2] + [None, None, None], None, None, None, None, None)
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # one record per step: ms_per_step_p50
    # The interpreter's cyclic garbage collector: a generation-2 pass over the ~10^6 objects torch has alive takes 50 - 70
    # ms, and when one lands inside a host-bound timed loop (the small shapes: 300 steps of 0.38 ms) it alone adds 0.13 -
    # 0.2 ms per step to the mean (measured: bench line 0.53 vs 0.385 ms/step at the reference's own shape; which run gets
    # one depends on how many container objects the set-up happened to allocate).  Everything alive now is moved to the
    # permanent generation, as a long-running training process does after start-up; the collector itself stays on.
    # BEFORE the warm-up steps (round 4): between them and the timed region it left the GPU idle for those 50 - 70 ms, the
    # clocks dropped, and a 20-step timed region (27 ms) ran its first steps on the way back up -- the driver's protocol
    # measured 5 % below a 200-step run of the same build on the same box.  Warm-up now ends where the timing starts.
    import gc
    gc.collect()
    gc.freeze()
    beat("warmup")
    run_steps(max(args.warmup - 1, 0))
    if args.warmup > 0 and not graph:
        # the last warm-up step carries stage events like every `every`-th timed step does: the first step that records
        # timing events pays a one-time 15 ms in the HIP runtime (measured), which is warm-up, not a step
        warm_ev = {-1: full_event_set()}
        run_steps(1, -1, warm_ev)
    elif args.warmup > 0:
        run_steps(1)
    # the time-tiled optimizer applies a row's zero-gradient updates late; what the WARM-UP steps still owe is applied here,
    # before the clock starts, as what the timed steps still owe is applied before it stops: the region pays for exactly its
    # own steps (round 4; before, it also paid the warm-up's share of the replay arithmetic)
    finish_adam()
    beat("timed")
    barrier()
    t0 = time.perf_counter()
    # (per-step marks cost the timed region 0.008 ms per step -- four alternating pairs at 20 steps, two at 200,
    #  profiles/r04_probes.md --: inside it only its first and last step are marked; the per-step distribution comes from a
    #  separate stretch of steps right behind it.  --step-marks-timed restores a mark per timed step)
    mark_every_step = args.step_marks_timed
    fb = run_steps(args.steps, 0, None if graph else events, step_marks)
    finish_adam()                  # inside the timed region: no update is left owing when the clock stops
    ev_tail = torch.cuda.Event(enable_timing=True)
    ev_tail.record()
    barrier()
    dt = time.perf_counter() - t0
    beat("after")
    steps_by_events_ms = step_marks[0].elapsed_time(step_marks[args.steps])
    if mark_every_step:
        per_step_ms = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(args.steps)]
        p50_what = "one HIP event per step inside the timed region (--step-marks-timed)"
    else:
        n_p = max(8, min(args.steps, 40))
        marks2 = [torch.cuda.Event(enable_timing=True) for _ in range(n_p + 1)]
        mark_every_step = True
        fb_p = run_steps(n_p, 0, None, marks2)
        fb = fb_p if fb is None else fb
        torch.cuda.synchronize()
        mark_every_step = False
        per_step_ms = [marks2[i].elapsed_time(marks2[i + 1]) for i in range(n_p)]
        p50_what = "one HIP event per step over %d steps right BEHIND the timed region (a mark per step costs it 0.008 ms/step)" % n_p
    flush_ms = step_marks[args.steps].elapsed_time(ev_tail)
    tiled = tiled and bool(inner._tiled_on())       # (a shard may have gone back to the sweep: HipBackend.note_requests)
    if graph:                      # stage timings from eager steps, outside the timed region
        model.enable_graph(False)
        graph = False
        events = {i: full_event_set() for i in range(0, 4 * every, every)}
        fb = run_steps(4 * every, 0, events)
        torch.cuda.synchronize()
        graph = True
    if not args.stage_events_timed and not graph:
        gather_ms = [v[0][0].elapsed_time(v[0][1]) for v in events.values()]      # live, inside the timed region
        # the stage table: eager steps with every boundary event, right behind the timed region (same state, same batches)
        n_st = max(3 * every, 12)
        events = {i: full_event_set() for i in range(0, n_st, every)}
        cm_ = getattr(model, "comm", None)
        if cm_ is not None and hasattr(cm_, "timing"):
            cm_.timing = True                 # (the per-rank table of the collectives: these steps, not the timed region)
        fb2 = run_steps(n_st, 0, events)
        if cm_ is not None and hasattr(cm_, "timing"):
            cm_.timing = False
        fb = fb2 if fb is None else fb
        torch.cuda.synchronize()
    else:
        gather_ms = [v[0][0].elapsed_time(v[0][1]) for v in events.values()]
    model.enable_stage_events(False)
    dt_ranks = [dt]
    ranks_seen = 1
    if dist is not None:
        # the step time of record is the MAX over ranks; every rank's own time rides in the same all_gather
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(got, mine)
        dt_ranks = [float(t.item()) for t in got]
        dt = max(dt_ranks)
    coll_table = None
    if dist is not None and hasattr(getattr(model, "comm", None), "timing_summary"):
        mine_t = {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in model.comm.timing_summary().items()}
        if ranks_seen > 1:
            coll_table = [None] * ranks_seen
            dist.all_gather_object(coll_table, mine_t)
        else:
            coll_table = [mine_t]
    if sharded:
        loss = float((fb[0][1] + args.reg_lambda * fb[0][2]).item())
    else:
        loss = float(fb[1][fb[0].loss].item())

    ev_sets = list(events.values())

    def avg(fn):
        return float(np.mean([fn(s) for s in ev_sets]))
    stages = {
        "fwd_gather_coattn": float(np.mean(gather_ms)),
        "fwd_gru": avg(lambda s: s[0][1].elapsed_time(s[0][2])),
        "fwd_attention": avg(lambda s: s[0][2].elapsed_time(s[0][3])),
        "fwd_head_loss": avg(lambda s: s[0][3].elapsed_time(s[0][4])),
        "bwd_head": avg(lambda s: s[1][0].elapsed_time(s[1][1])),
        "bwd_attention": avg(lambda s: s[1][1].elapsed_time(s[1][2])),
        "bwd_gru": avg(lambda s: s[1][2].elapsed_time(s[1][3])),
        "bwd_coattn_scatter": avg(lambda s: s[1][3].elapsed_time(s[1][4])),
        "bwd_weight_grads": avg(lambda s: s[1][4].elapsed_time(s[1][5])),
        "adam_table_and_dense": avg(lambda s: s[2].elapsed_time(s[3])),
    }
    if tiled:      # the rows of the batch brought up to date before the forward (outside every stage above)
        vals = []
        for s_ in ev_sets:      # (the sharded path fetches a step ahead: a step may contain no catch-up at all)
            try:
                vals.append(s_[4].elapsed_time(s_[5]))
            except Exception:
                pass
        stages["adam_catchup_batch_rows"] = float(np.mean(vals)) if vals else None
    A = int(getattr(batches[0], "active_slices", 0)) or T      # slices the gather really reads
    ev_overhead_ms = event_pair_overhead_ms()
    ab, R = alg_bytes_per_sample(A, K, D, Fu, Fi)
    ab_full, _ = alg_bytes_per_sample(T, K, D, Fu, Fi)
    gather_s = stages["fwd_gather_coattn"] * 1e-3
    achieved = ab * B / gather_s / 1e9
    N = kw["feature_size"]
    n_w = model.n_w
    live_rows = int((inner.table_flags > 0).sum().item())
    rows_local = int(inner.table.shape[0])
    if sharded:
        touched = min(live_rows, R * B)               # upper bound (not counted on the sharded path)
    else:                                             # one extra untimed backward: count the rows it marks
        model.forward_backward(batches[0], args.reg_lambda, 0.8)
        touched = int((inner.table_flags == 2).sum().item())
        inner._drop_row_marks()
    if tiled:   # score_adam_touched: p, m, v read and written + g read on the rows with a gradient; state-byte scan; dense vars
        adam_bytes = 4 * D * 7 * touched + rows_local + 7 * 4 * n_w
        adam_name = ("adam_touched (time-tiled optimizer: 7 fp32 streams over the rows with a gradient, the state-byte scan, "
                     "+ dense vars; the owed zero-gradient updates run in adam_catchup_batch_rows and, beside the step, in "
                     "score_adam_catchup_rows)")
    else:
        adam_bytes = 4 * D * (6 * live_rows + touched) + rows_local + 7 * 4 * n_w
        adam_name = "adam_rows (6 fp32 streams over the live table rows, + g on touched rows, + dense vars)"
    adam_timed = stages["adam_table_and_dense"] > 1e-3        # (pipelined sharded step: the update runs inside the step)
    if not adam_timed:
        stages["adam_table_and_dense"] = None
    scat_bytes = R * (4 + 4 * D) * B
    headline_live_frac = live_rows / float(rows_local)

    # ---------------------------------------------------------------- the reference's own call, synchronous
    # loss = model.train(sess, batch_data, lr, reg_lambda) (score.py:101-116: one sess.run per step, the loss read back
    # every step) on the same device-resident batches and the same state as the headline: the host waits for every step,
    # so the queue drains each time and nothing of step t+1 is enqueued under step t
    beat("sync_train")
    n_sync = max(10, min(args.steps, 100))
    sync_losses = []

    def sync_steps(n):
        for i in range(n):
            beat()
            b_ = batches[i % len(batches)]
            if sharded:
                nxt_ = batches[(i + 1) % len(batches)] if (args.prefetch and i + 1 < n) else None
                sync_losses.append(model.train(None, b_, args.lr, args.reg_lambda, next_batch=nxt_))
            else:
                sync_losses.append(model.train(None, b_, args.lr, args.reg_lambda))
    sync_steps(2)
    barrier()
    t_sync = time.perf_counter()
    sync_steps(n_sync)
    finish_adam()
    barrier()
    dt_sync = time.perf_counter() - t_sync
    if dist is not None:
        t_ = torch.tensor([dt_sync], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        dt_sync = float(t_.item())
    beat("side")

    # ---------------------------------------------------------------- side measurements (one GPU, unsharded)
    side = {}
    do_side = world_size == 1 and not sharded and not args.no_side and not args.fresh_state and not args.no_skip_masked
    if do_side:
        k2 = max(20, min(args.steps, 100))

        def timed(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            run_steps(n)
            finish_adam()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        # (a) all time slices computed, every row live
        model.skip_masked_slices = False
        keep = batches
        batches = [model.device_batch(tuple(db.tensors)) for db in keep]       # same tensors, active_slices = 0
        assert all(db.active_slices == 0 for db in batches)
        run_steps(3)
        s_all = timed(k2)
        side["value_all_slices"] = B / s_all
        model.skip_masked_slices = True
        batches = keep
        # (b) best case: fresh optimizer state -- only rows touched during the run carry moments
        for t_ in (model.table_m, model.table_v, model.w_m, model.w_v):
            t_.zero_()
        model.table_flags.zero_()
        run_steps(args.warmup)
        s_best = timed(k2)
        side["value_best_case"] = B / s_best
        side["best_case_live_row_frac"] = int((model.table_flags > 0).sum().item()) / float(rows_local)
        model.table_flags.fill_(1)
        if tiled:
            # (b2) the per-step dense sweep the time-tiled optimizer replaces (bit-identical results), same state
            w_ = inner.adam_window
            inner.adam_window = 0
            run_steps(3)
            side["value_dense_adam_sweep"] = B / timed(k2)
            inner.adam_window = w_
        # (c) low-duplication gather probe
        try:
            pr = gather_probe(model, kw, B, args.probe_rows, 20)
            p_tr, p_src = committed_traffic(args.config, "gather_probe", "coattn_fwd_kernel")
            s_ = pr["ms"] * 1e-3
            side["roofline_lowdup"] = {
                "kernel": "coattn_fwd_kernel on a low-duplication batch: uniform ids over a %d-row probe table (%.1f GB), no "
                          "dummy slices, nothing shared between candidates, all %d slices" % (pr["probe_rows"],
                                                                                        pr["probe_rows"] * D * 4 / 1e9, T),
                "bound": "hbm", "achieved": pr["alg_bytes"] / s_ / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": pr["alg_bytes"] / s_ / 1e9 / HBM_PEAK_GBS, "traffic": p_tr, "traffic_source": p_src,
                "frac_on_traffic": (p_tr / s_ / 1e9 / HBM_PEAK_GBS) if p_tr else None,
                "algorithmic_bytes_per_launch": pr["alg_bytes"], "avg_launch_ms": pr["ms"], "min_launch_ms": pr["ms_min"],
                "avg_launch_ms_net_of_event_overhead": pr["ms"] - ev_overhead_ms,
                "expected_distinct_rows": pr["expected_distinct_rows"], "row_uses": pr["row_uses"],
                "compulsory_row_bytes": pr["compulsory_row_bytes"]}
        except Exception as e:          # an optional leg never takes the headline down
            side["roofline_lowdup"] = {"error": repr(e)}
        # (c2) the largest matrix product of the forward pass, alone
        try:
            gp = panel_gemm_probe(model, kw, B, A)
            if gp is not None:
                g_tr, g_src = committed_traffic(args.config, "bench_workload", "gemm_panel_kernel", sources=("gemm_panel.hip",))
                s_ = gp["ms"] * 1e-3
                side["roofline_gemm_panel"] = {
                    "kernel": "gemm_panel_kernel: the GRU input projections of both sides in one launch, 2 x [%d, %d] . [%d, %d] + "
                              "bias, fp32-accurate on the bf16 matrix cores (bf16x3: six v_mfma_f32_16x16x32_bf16 per product)"
                              % (gp["M"], gp["K"], gp["K"], gp["N"]),
                    "bound": "mfma", "achieved": 6 * gp["flops"] / s_ / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6 * gp["flops"] / s_ / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "fp32_equivalent_tflops": gp["flops"] / s_ / 1e12,
                    "algorithmic_bytes_per_launch": gp["a_bytes"] + gp["c_bytes"] + gp["w_bytes"],
                    "traffic": g_tr, "traffic_source": g_src, "avg_launch_ms": gp["ms"], "min_launch_ms": gp["ms_min"],
                    "max_rel_err_vs_fp64": gp["max_rel_err_vs_fp64"],
                    "note": "achieved counts the six bf16 MFMAs each fp32-accurate product costs (the work the pipe does); "
                            "fp32_equivalent_tflops counts the product once.  Measured alone on synthetic operands of the bench "
                            "workload's shape through score_gemm_panel_run (include/score_hip.h); in the step it is the first "
                            "kernel of the fwd_gru stage"}
        except Exception as e:
            side["roofline_gemm_panel"] = {"error": repr(e)}
        # (d) host ingestion included
        try:
            from score_amd.synth import make_graph
            from score_amd.graph import DeviceGraphLoader
            au, ai = min(world.U, 50000), min(world.I, 200000)
            g = make_graph(world, T + 1, active_users=au, active_items=ai).to_device(model.device)
            rng = np.random.default_rng(5)
            n_lines = (B // 2) * 60
            lines = list(zip(rng.integers(1, au + 1, n_lines).tolist(),
                             np.stack([rng.integers(world.U + 1, world.U + ai + 1, n_lines),
                                       rng.integers(world.U + 1, world.U + ai + 1, n_lines)], 1).tolist()))
            loader = DeviceGraphLoader(g, B, lines, 0, T - 2, 1, T, K)
            it = iter(loader)
            for _ in range(5):
                model.train_async(next(it), args.lr, args.reg_lambda)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n_ing = 0
            for b_ in it:
                model.train_async(b_, args.lr, args.reg_lambda)
                n_ing += 1
            torch.cuda.synchronize()
            s_ing = (time.perf_counter() - t) / n_ing
            nested3 = [world.batch(B, 77 + i, as_lists=True) for i in range(3)]
            nested = nested3[0]
            model.train(None, nested, args.lr, args.reg_lambda)
            t = time.perf_counter()
            for i in range(6):
                model.train(None, nested3[i % 3], args.lr, args.reg_lambda)
            s_nested = (time.perf_counter() - t) / 6
            # the same tuples through SCOREBASE.feed: converted and uploaded one or two batches ahead on a worker thread
            n_feed = 24
            t = time.perf_counter()
            for db_ in model.feed((nested3[i % 3] for i in range(n_feed))):
                model.train(None, db_, args.lr, args.reg_lambda)
            s_feed = (time.perf_counter() - t) / n_feed
            ft = model.feed_threads
            model.feed_threads = 1
            t = time.perf_counter()
            for i in range(3):
                model.train(None, nested3[i % 3], args.lr, args.reg_lambda)
            s_nested_1t = (time.perf_counter() - t) / 3
            model.feed_threads = ft
            del nested3
            side["ingestion"] = {
                "device_assembly_samples_per_s": B / s_ing, "device_assembly_ms_per_step": s_ing * 1e3, "steps": n_ing,
                "what": "DeviceGraphLoader (score_batch_assemble: CSR graph in HBM -> the eight int32 tensors, one launch per "
                        "batch) + SCORE.train_async inside the timed loop; synthetic graph over the config's id space",
                "nested_python_lists_samples_per_s": B / s_nested, "nested_python_lists_ms_per_step": s_nested * 1e3,
                "nested_what": "model.train(sess, batch_data, ...) fed the 8-tuple of nested Python lists exactly as "
                               "GraphLoader yields it (graph_loader.py:383), one synchronous call after the other: list -> "
                               "int32 conversion (%d native threads without the GIL, pinned staging) + H2D copy + step + "
                               "loss read-back; host-bound, never `value`" % ft,
                "nested_python_lists_feed_ahead_samples_per_s": B / s_feed,
                "nested_python_lists_feed_ahead_ms_per_step": s_feed * 1e3,
                "feed_ahead_what": "for b in model.feed(loader): model.train(sess, b, ...) -- the same tuples converted and "
                                   "uploaded by a worker thread one or two batches ahead, under the running step",
                "nested_python_lists_single_thread_samples_per_s": B / s_nested_1t, "feed_threads": ft}
            del g, loader
        except Exception as e:
            side["ingestion"] = {"error": repr(e)}

    if rank != 0:
        if hb is not None:
            hb.stop()
        if dist is not None:
            dist.destroy_process_group()
        return
    beat("report")
    try:
        ceiling = stream_copy_ceiling(inner)
    except Exception as e:          # (a measurement aid never takes the headline down)
        ceiling = {"error": repr(e)}
    traffic, tsrc = committed_traffic(args.config, "bench_workload", "coattn_fwd_kernel")
    scat_tr, scat_src = committed_traffic(args.config, "bench_workload",
                                          ("coattn_bwd_kernel_t", "pull_kernel", "pull_fixup_kernel", "pull_long_kernel",
                                           "target_bwd_kernel"), ("embed.hip", "scatter.hip"))
    adam_tr, adam_src = committed_traffic(args.config, "bench_workload",
                                          ("adam_step_kernel", "adam_touched_kernel", "adam_rows_kernel", "adam_kernel"), ("adam_tiled.hip", "head.hip"))
    bench_block = {"kernel": "coattn_fwd_kernel (fused embedding gather + co-attention, both calls, one launch) on the bench "
                             "workload",
                   "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                   "frac_on_traffic": (traffic / gather_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                   "algorithmic_bytes_per_launch": ab * B, "avg_launch_ms": stages["fwd_gather_coattn"],
                   "note": "achieved/frac count the ALGORITHMIC bytes of SURVEY 8(d) (every row use); the loader-shaped batch "
                           "repeats hot rows, the dummy row and the user side of both candidates, which L1/L2/Infinity Cache "
                           "serve, so this frac is NOT an HBM utilisation and can exceed 1: frac_on_traffic (PMC bytes of a "
                           "committed profile of this very build, else null) is.  The block named `roofline` is the same "
                           "kernel where algorithmic bytes ~ traffic",
                   "event_pair_overhead_ms": ev_overhead_ms,
                   "avg_launch_ms_net_of_event_overhead": stages["fwd_gather_coattn"] - ev_overhead_ms,
                   "time_slices_gathered": A, "time_slices_fed": T,
                   "algorithmic_bytes_per_launch_if_all_fed_slices_were_gathered": ab_full * B}
    lowdup = side.pop("roofline_lowdup", None)
    if lowdup is not None and "error" not in lowdup:
        roofline = dict(lowdup)
        roofline["note"] = ("the dominant HBM-bound kernel of the path, measured where its algorithmic bytes (SURVEY 8(d): every "
                            "row use + ids + reduced outputs) equal what memory has to deliver: ids uniform over a probe table "
                            "far beyond the 256 MiB Infinity Cache.  avg_launch_ms: HIP events on the launch stream, live in "
                            "this run; traffic: PMC bytes of the committed profile of this build of embed.hip (null if the "
                            "profile is stale).  The same kernel on the loader-shaped bench batches: roofline_bench_workload")
    else:
        roofline = dict(bench_block)
        roofline["note"] = ("low-duplication probe not run in this invocation (%s): this is the loader-shaped bench workload, "
                            "whose algorithmic bytes exceed its memory traffic -- see roofline_bench_workload.note"
                            % ("--no-side / sharded / non-default state" if lowdup is None else lowdup.get("error")))
    if "GBs" in ceiling:     # both denominators -- the spec peak and the copy ceiling measured in this run
        for blk in (roofline, bench_block):
            blk["peak_measured"] = ceiling["GBs"]
            blk["frac_of_peak_measured"] = blk["achieved"] / ceiling["GBs"]
        roofline["peak_measured_what"] = ceiling["what"] + " (%.3f ms per launch)" % ceiling["ms"]
    else:
        roofline["peak_measured"] = None
        roofline["peak_measured_error"] = ceiling.get("error")
    out = {
        "metric": "train samples/sec @ batch=1024",
        "value": B * world_size * args.steps / dt,
        "unit": "samples/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_p50": float(np.median(per_step_ms)),
        "ms_per_step_p10_p90": [float(np.percentile(per_step_ms, 10)), float(np.percentile(per_step_ms, 90))],
        "timed_region_ms": {"wall": dt * 1e3, "steps_by_events": steps_by_events_ms, "optimizer_flush_by_events": flush_ms,
                            "first_steps_of_the_marked_stretch": [round(x, 4) for x in per_step_ms[:6]]},
        "ms_per_step_what": "ms_per_step = wall clock of the timed region (barrier to barrier, the optimizer's flush included) "
                            "/ steps: the figure `value` is computed from.  p50 / p10 / p90: per-step durations, " + p50_what + " (rank 0)",
        "value_sync_train": B * world_size * n_sync / dt_sync,
        "value_sync_train_what": "the reference's own call, one after the other: loss = model.train(sess, batch_data, lr, "
                                 "reg_lambda) (score.py:101-116) with its per-step loss read-back, %d steps on the same "
                                 "device-resident batches and optimizer state as `value` (which enqueues forward_backward + "
                                 "apply_adam without reading the loss back: the asynchronous form); ms_per_step %.4f, last loss %.6f"
                                 % (n_sync, dt_sync / n_sync * 1e3, sync_losses[-1]),
        "ms_per_step_ranks": {"min": min(dt_ranks) / args.steps * 1e3, "max": max(dt_ranks) / args.steps * 1e3},
        "rccl_ranks": ranks_seen if (dist is not None and backend == "nccl") else 0,
        "dist": ({"backend": backend, "world_size": ranks_seen, "a2a_probe": getattr(getattr(model, "comm", None), "a2a_probe", None),
                  "collectives_ms": coll_table,
                  "collectives_what": "per rank, averaged over the dozen stage-table steps right behind the timed region (not inside it: the events cost host time), a pair of HIP events around each data-path collective on "
                                      "the stream it is issued on: a2a_row_requests (int32 ids), a2a_rows (fp32 rows back), a2a_row_grads "
                                      "(fp32 row gradients to the owners), all_reduce (flat dense gradient + the loss slot) -- what the "
                                      "first multi-GPU curve is read against DESIGN section 5's prediction with"}
                 if dist is not None else None),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "parity": "partial: the CPU oracle the tests check against restates a TF-1.x graph; the reference ships no golden "
                  "vectors and TensorFlow cannot run here, so the oracle itself is unpinned (DESIGN.md section 2)",
        "launch": ("one captured hipGraph per step (SCOREBASE.enable_graph); stages_ms from eager steps outside the timed loop"
                   if args.graph else "eager launches"),
        "config": {"workload": "%s: SCORE full train step (fwd + bwd + dense TF-Adam), N=%d rows, T=%d, K=%d, "
                               "D=%d, H=%d, Fu=%d, Fi=%d, per-GPU batch %d (global %d), keep_prob 0.8, "
                               "%d distinct pre-staged batches, length=%d for every sample (slices >= length are "
                               "masked by the model and skipped)" % (args.config, N, T, K, D, kw["hidden_size"], Fu, Fi,
                                                                   B, B * world_size, len(batches), A),
                   "optimizer_state": ("fresh (best case): only rows touched during the run carry Adam moments"
                                       if args.fresh_state else
                                       "steady state: every table row carries Adam moments (live_row_frac 1.0), so dense "
                                       "ApplyAdam owes every row an update every step"),
                   "table_optimizer": (("time-tiled ApplyAdam, window %d (include/score_hip.h: the zero-gradient update of a "
                                        "row is applied when the row is next read, or once per window, in step order: "
                                        "bit-identical to the per-step sweep, tests/test_gpu_adam_tiled.py; every update "
                                        "still owed when the timed steps end is applied INSIDE the timed region); "
                                        "value_dense_adam_sweep is the same run with the per-step sweep") % inner.adam_window
                                       if tiled else "dense ApplyAdam sweep over the live rows every step"),
                   "live_row_frac": headline_live_frac,
                   "table": "row-sharded row%%G over %d GPU(s)" % world_size if world_size > 1 else "single GPU",
                   "final_loss": loss},
        "roofline": roofline,
        "roofline_bench_workload": bench_block,
        "roofline_other": {
            **({"gru input projections (gemm_panel_kernel)": side.pop("roofline_gemm_panel")} if "roofline_gemm_panel" in side else {}),
            adam_name: {
                "live_row_frac": headline_live_frac, "rows_with_gradient_per_step": touched,
                "bound": "hbm", "achieved": adam_bytes / (stages["adam_table_and_dense"] * 1e-3) / 1e9 if adam_timed else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": adam_bytes / (stages["adam_table_and_dense"] * 1e-3) / 1e9 / HBM_PEAK_GBS if adam_timed else None,
                "algorithmic_bytes_per_step": adam_bytes, "traffic": adam_tr, "traffic_source": adam_src,
                "frac_on_traffic": (adam_tr / (stages["adam_table_and_dense"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                if (adam_tr and adam_timed) else None},
            "coattn_bwd + scatter (R*(4+4D) per sample, algorithmic)": {
                "bound": "hbm", "achieved": scat_bytes / (stages["bwd_coattn_scatter"] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": scat_bytes / (stages["bwd_coattn_scatter"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": scat_bytes, "traffic": scat_tr, "traffic_source": scat_src,
                "frac_on_traffic": (scat_tr / (stages["bwd_coattn_scatter"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if scat_tr else None}},
        "stages_ms": stages,
    }
    out.update(side)
    small = None
    if world_size == 1 and not args.no_cpu_baseline:
        params = model.get_params()
        del model, batches, inner
        torch.cuda.empty_cache()
        if do_side and not args.no_small_shapes:
            # before the CPU legs (their threads would compete with this thread's launch calls), printed behind them
            # each leg in a CHILD process of its own (started, not exec'ed: this process keeps the GPU): in this one the headline's
            # streams are still alive and the new model's four streams land on hardware queues they share with them -- two streams
            # on one queue serialise (round 6: Tmall default 0.205 ms here against 0.189 in a fresh process on the same box)
            import subprocess
            small = {}
            for c in ("cfg2", "tmall_default"):
                try:
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--small-shape-leg", c, "--lr", str(args.lr),
                                         "--reg-lambda", str(args.reg_lambda)], capture_output=True, text=True, timeout=300,
                                        cwd=ROOT)
                    lines = [l for l in cp.stdout.strip().splitlines() if l.strip().startswith("{")]
                    small[c] = json.loads(lines[-1]) if (cp.returncode == 0 and lines) else {"error": (cp.stderr or "no output")[-200:]}
                except Exception as e:          # an optional leg never takes the headline down
                    small[c] = {"error": repr(e)[:200]}
            small["what"] = ("model.train_async in a fresh process per leg, device-resident batches, next batch announced, every row "
                             "live; ms_per_step = wall incl. optimizer flush; host_us = enqueue loop only; launches from "
                             "profiles/%s_<config>_sequence.txt" % PROFILE_ROUND)
        out["cpu_baseline"] = cpu_baseline(kw, world.batch(B, 1000), params)
        if do_side:
            # the literal materialised-tile form at the reference's own Tmall-default shape (BASELINE.md section 3;
            # 11 GB per batch at cfg-3: infeasible beyond)
            from oracle import score_oracle as so
            w2, kw2 = make_world("tmall_default")
            B2 = kw2.pop("batch")
            cfg2 = so.Cfg(kw2["feature_size"], kw2["eb_dim"], kw2["hidden_size"], kw2["max_time_len"],
                          kw2["obj_per_time_slice"], kw2["user_fnum"], kw2["item_fnum"], "SCORE")
            P2 = so.init_params(cfg2, 3)
            out["cpu_baseline_literal_tile"] = cpu_baseline(kw2, w2.batch(B2, 5), P2, budget_s=12.0, max_steps=3, tiled=True)
            out["cpu_baseline_tmall_default_collapsed"] = cpu_baseline(kw2, w2.batch(B2, 5), so.init_params(cfg2, 3),
                                                                       budget_s=8.0, max_steps=3)
    else:
        out["cpu_baseline"] = None
    if hb is not None:
        hb.stop()
    if dist is not None:
        dist.destroy_process_group()
    if small is not None:
        # LAST key of the line: the driver's record keeps the line's last 2,000 characters
        out["small_shapes"] = small
    # RCCL prints its version banner through C stdio: flush it first so the JSON is the LAST line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
