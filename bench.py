#!/usr/bin/env python3
"""bench.py -- train samples/sec of the SCoRe hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one full training step of SCORE (score.py:101-116: forward, backward,
dense TF-Adam over the whole table and all dense variables) on one synthetic
Tmall-shaped batch whose int32 index tensors are already resident in HBM.
Prints ONE JSON line (rank 0).  `roofline` describes the fused gather + co-attention
forward kernel (the embedding-gather kernel BASELINE.json's north_star targets):
algorithmic bytes per launch / its average duration measured with HIP
events recorded on the launch stream inside the timed region.  `cpu_baseline` times
the CPU restatement of the TF graph (oracle/, "port") on the host cores for a bounded
number of steps of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def cpu_baseline(kw, world, B, params, budget_s=25.0, max_steps=4, max_threads=32):
    """CPU restatement of the TF1 graph (Oracle B + dense TF-Adam), full train step, on a bounded
    sample: the first step is timed too and is the whole sample if it alone exceeds the budget."""
    from oracle import score_oracle as so
    threads = min(usable_cpus(), max_threads)
    torch.set_num_threads(threads)
    m = so.OracleModel(kw["feature_size"], kw["eb_dim"], kw["hidden_size"], kw["max_time_len"],
                       kw["obj_per_time_slice"], kw["user_fnum"], kw["item_fnum"], "SCORE", params=params)
    batch = world.batch(B, 1000)
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
    return {"value": B / med, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d full train step(s) (fwd+bwd+dense TF-Adam) of the same workload after one %s step, B=%d, "
                      "median %.3f s/step; CPU restatement of the TF1 graph (TensorFlow unavailable), torch-CPU "
                      "fp32, %d threads (os.cpu_count()=%d, usable=%d)"
                      % (max(len(times), 1), "untimed warm-up" if times else "(timed, no warm-up)", B, med, threads,
                         os.cpu_count(), usable_cpus())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batches", type=int, default=4, help="distinct pre-staged batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded all-to-all path even with one rank (exercises RCCL plumbing)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="sharded path: do NOT start the next batch's index plan under this step's compute "
                         "(default on: one rank through RCCL measured 2.31 vs 2.55 ms/step)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="sharded path: optimizer after the backward pass instead of the pipelined step (table update "
                         "and the next batch's row fetch under this step's weight-gradient tail and dense all-reduce)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the HIP stage events (live kernel timing for `roofline`) on every E-th timed step: "
                         "eleven timing events per step cost ~3 %% of a 1.8 ms step")
    ap.add_argument("--all-rows-live", action="store_true",
                    help="mark every table row as carrying Adam moments before the run: the long-run state of "
                         "dense Adam (its sweep then moves 6 fp32 streams over the whole table every step)")
    ap.add_argument("--no-skip-masked", action="store_true",
                    help="gather and compute all T time slices, also those past every sample's length (whose "
                         "results the model masks): A/B for score_batch_t.active_slices")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--reg-lambda", type=float, default=1e-4)
    args = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # rehearsal aids (several ranks on a one-GPU box): SCORE_BENCH_DEVICE pins every rank to one device,
    # SCORE_DIST_BACKEND=gloo swaps RCCL for gloo (device tensors staged through host memory, score_amd/dist.py)
    if os.environ.get("SCORE_BENCH_DEVICE"):
        local_rank = int(os.environ["SCORE_BENCH_DEVICE"])
    backend = os.environ.get("SCORE_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world_size > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)

    from score_amd.synth import make_world
    from score_amd.model import SCORE
    world, kw = make_world(args.config)
    B = kw.pop("batch")
    T, K, D = kw["max_time_len"], kw["obj_per_time_slice"], kw["eb_dim"]
    Fu, Fi = kw["user_fnum"], kw["item_fnum"]

    if sharded:
        from score_amd.dist import ShardedSCORE
        model = ShardedSCORE(seed=1111, **kw)
    else:
        model = SCORE(seed=1111, **kw)
    if args.no_skip_masked:
        (model.backend.m if sharded else model).skip_masked_slices = False
    # weak scaling: every rank trains on its own B-sample batches (global batch = B * N)
    batches = [model.device_batch(world.batch(B, rank * 1000 + i)) for i in range(args.batches)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    inner = model.backend.m if sharded else model     # owns the table (shard) and its optimizer state
    if args.all_rows_live:
        inner.table_flags.fill_(1)
    for i in range(args.warmup):
        if sharded and args.prefetch:
            model.train_async(batches[i % len(batches)], args.lr, args.reg_lambda,
                              next_batch=batches[(i + 1) % len(batches)])
        else:
            model.train_async(batches[i % len(batches)], args.lr, args.reg_lambda)
    # per-step stage events for the live kernel timing
    ev_sets, ev_at = [], {}
    every = max(1, args.event_every)
    for i in range(args.steps):
        if i % every:
            continue
        model.enable_stage_events(True)
        ev_at[i] = len(ev_sets)
        ev_sets.append((model.fwd_events, model.bwd_events, torch.cuda.Event(enable_timing=True),
                        torch.cuda.Event(enable_timing=True)))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in ev_at:
            model.fwd_events, model.bwd_events, e_a0, e_a1 = ev_sets[ev_at[i]]
        else:
            model.fwd_events = model.bwd_events = e_a0 = e_a1 = None
        if sharded:   # optionally run the next batch's index-only phase (plan + row requests) inside this step
            nxt = batches[(i + 1) % len(batches)] if (args.prefetch and i + 1 < args.steps) else None
            # with a next batch the step is pipelined: the optimizer runs inside (apply_adam below is then a no-op)
            fb = model.forward_backward(batches[i % len(batches)], args.reg_lambda, 0.8, None, nxt,
                                        lr=args.lr if args.pipeline else None)
        else:
            fb = model.forward_backward(batches[i % len(batches)], args.reg_lambda, 0.8)
        if e_a0 is not None:
            e_a0.record()
        model.apply_adam(args.lr, args.reg_lambda)
        if e_a1 is not None:
            e_a1.record()
    barrier()
    dt = time.perf_counter() - t0
    model.enable_stage_events(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if sharded:
        loss = float((fb[0][1] + args.reg_lambda * fb[0][2]).item())
    else:
        loss = float(fb[1][fb[0].loss].item())

    # live stage timings (ms), averaged over the timed steps
    def avg(fn):
        return float(np.mean([fn(s) for s in ev_sets]))
    stages = {
        "fwd_gather_coattn": avg(lambda s: s[0][0].elapsed_time(s[0][1])),
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
    # slices the gather really reads: synthetic batches have length = T-2 (the reference's train split has 9 of
    # 11, graph_loader.py:382) and the slices past the longest sample, which the model masks out of every
    # result, are skipped (score_batch_t.active_slices) -- the numerator counts only what is gathered
    A = int(getattr(batches[0], "active_slices", 0)) or T
    # what an EMPTY pair of timing events measures on this stream (two marker packets back to back): the stage
    # durations above include it, rocprofv3's per-kernel durations do not
    cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    pad = torch.zeros((1 << 20,), device="cuda")
    for a_, b_ in cal:
        pad.add_(1.0)
        a_.record()
        b_.record()
    torch.cuda.synchronize()
    ev_overhead_ms = float(np.median([a_.elapsed_time(b_) for a_, b_ in cal]))
    ab, R = alg_bytes_per_sample(A, K, D, Fu, Fi)
    ab_full, _ = alg_bytes_per_sample(T, K, D, Fu, Fi)
    gather_s = stages["fwd_gather_coattn"] * 1e-3
    achieved = ab * B / gather_s / 1e9
    N = kw["feature_size"]
    n_w = model.n_w
    # dense ApplyAdam driven by the row state bytes: live rows move p, m, v in and out (6 streams), rows with
    # a gradient this step read it too, every row costs its state byte
    live_rows = int((inner.table_flags > 0).sum().item())
    rows_local = int(inner.table.shape[0])
    if sharded:
        touched = min(live_rows, R * B)               # upper bound (not counted on the sharded path)
    else:                                             # one extra untimed backward: count the rows it marks
        model.forward_backward(batches[0], args.reg_lambda, 0.8)
        touched = int((inner.table_flags == 2).sum().item())
    adam_bytes = 4 * D * (6 * live_rows + touched) + rows_local + 7 * 4 * n_w
    adam_timed = stages["adam_table_and_dense"] > 1e-3        # (pipelined sharded step: the update runs inside the step)
    if not adam_timed:
        stages["adam_table_and_dense"] = None
    scat_bytes = R * (4 + 4 * D) * B

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    # HBM bytes per launch of the gather kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # --pmc WRITE_SIZE, separate runs, gfx950 FETCH_SIZE x2 correction; tools/summarize_profile.py)
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r01_%s_pmc_traffic.json" % args.config)))
        traffic = [v["hbm_bytes"] for k, v in pj["kernels"].items() if "coattn_fwd_kernel" in k][0]
    except Exception:
        traffic = None
    out = {
        "metric": "train samples/sec @ batch=1024",
        "value": B * world_size * args.steps / dt,
        "unit": "samples/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: SCORE full train step (fwd + bwd + dense TF-Adam), N=%d rows, T=%d, K=%d, "
                               "D=%d, H=%d, Fu=%d, Fi=%d, per-GPU batch %d (global %d), keep_prob 0.8, "
                               "%d distinct pre-staged batches, length=%d for every sample (slices >= length are "
                               "masked by the model and skipped)" % (args.config, N, T, K, D, kw["hidden_size"], Fu, Fi,
                                                                   B, B * world_size, len(batches), A),
                   "table": "row-sharded row%%G over %d GPU(s)" % world_size if world_size > 1 else "single GPU",
                   "final_loss": loss},
        "roofline": {"kernel": "coattn_fwd_kernel (fused embedding gather + co-attention, both calls, one launch)",
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ab * B, "avg_launch_ms": stages["fwd_gather_coattn"],
                     "hbm_frac_measured_traffic": (traffic / gather_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "note": "achieved/frac use the ALGORITHMIC bytes of SURVEY 8(d) (every row use counted); about three "
                             "quarters of them are repeats of hot rows and of the dummy row that L1/L2/Infinity Cache serve "
                             "(traffic = HBM bytes measured with PMC counters), so frac can exceed 1: it says how close the "
                             "kernel is to what HBM could deliver if every byte came from it.  What bounds it is the request "
                             "latency of the L2-miss path x waves in flight (profiles/r01_cfg3_gather_counters.md)",
                     "event_pair_overhead_ms": ev_overhead_ms,
                     "avg_launch_ms_net_of_event_overhead": stages["fwd_gather_coattn"] - ev_overhead_ms,
                     "time_slices_gathered": A, "time_slices_fed": T,
                     "algorithmic_bytes_per_launch_if_all_fed_slices_were_gathered": ab_full * B},
        "roofline_other": {
            "adam_rows (6 fp32 streams over the live table rows, + g on touched rows, + dense vars)": {
                "live_row_frac": live_rows / float(rows_local), "rows_with_gradient_per_step": touched,
                "bound": "hbm", "achieved": adam_bytes / (stages["adam_table_and_dense"] * 1e-3) / 1e9 if adam_timed else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": adam_bytes / (stages["adam_table_and_dense"] * 1e-3) / 1e9 / HBM_PEAK_GBS if adam_timed else None},
            "coattn_bwd + scatter (R*(4+4D) per sample)": {
                "bound": "hbm", "achieved": scat_bytes / (stages["bwd_coattn_scatter"] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": scat_bytes / (stages["bwd_coattn_scatter"] * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        "stages_ms": stages,
    }
    if world_size == 1 and not args.no_cpu_baseline:
        params = model.get_params()
        del model, batches
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(kw, world, B, params)
    else:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.destroy_process_group()
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
