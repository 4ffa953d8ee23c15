"""bench.py's output contract: ONE JSON line with the driver's keys, the `roofline` and `cpu_baseline` objects, and the
round-2 companions of the steady-state headline (a short run; the CPU leg is bounded by its own budget)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, **env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    return json.loads(lines[-1])           # the JSON is the LAST line


def test_default_line_carries_the_contract():
    j = _run("--steps", "12", "--warmup", "3", "--config", "cfg2", "--probe-rows", "2000000")
    for k, v in (("metric", "train samples/sec @ batch=1024"), ("unit", "samples/s"), ("n_gpus", 1), ("steps", 12), ("warmup", 3),
                 ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None), ("dtype", "f32"), ("data", "synthetic")):
        assert j[k] == v, (k, j[k])
    assert j["value"] > 0 and abs(j["value"] - 256 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["live_row_frac"] == 1.0
    for key in ("roofline", "roofline_bench_workload"):
        r = j[key]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert "traffic" in r and r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0
    assert abs(j["roofline"]["achieved"] - j["roofline"]["algorithmic_bytes_per_launch"] / (j["roofline"]["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * j["roofline"]["achieved"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # the headline's companions
    assert j["value_best_case"] > 0 and j["value_all_slices"] > 0 and 0 < j["best_case_live_row_frac"] <= 1
    lo = j["roofline"]                  # the headline block is the low-duplication probe
    assert "low-duplication" in lo["kernel"] and lo["frac"] <= 1.0 and lo["expected_distinct_rows"] <= lo["row_uses"]
    for v in j["roofline_other"].values():
        assert "traffic" in v and (v.get("algorithmic_bytes_per_step") or v.get("algorithmic_bytes_per_launch")) > 0
    assert "gru input projections (gemm_panel_kernel)" not in j["roofline_other"]      # (cfg-2's 96 columns: tiled kernels)
    ing = j["ingestion"]
    assert ing["device_assembly_samples_per_s"] > 0 and ing["nested_python_lists_samples_per_s"] > 0
    assert j["cpu_baseline_literal_tile"]["value"] > 0 and "materialised" in j["cpu_baseline_literal_tile"]["sample"]
    assert set(j["stages_ms"]) >= {"fwd_gather_coattn", "bwd_coattn_scatter", "adam_table_and_dense"}
    # the small shapes ride in the default line, as its LAST key (the driver keeps the line's last 2,000 characters)
    assert list(j)[-1] == "small_shapes"
    for c in ("cfg2", "tmall_default"):
        leg = j["small_shapes"][c]
        assert set(leg) >= {"samples_per_s", "ms_per_step", "ms_p50", "host_us_per_step", "launches_per_step", "B", "steps", "warmup", "form"}, leg
        assert leg["samples_per_s"] > 0 and leg["form"] == "per-sample" and abs(leg["samples_per_s"] - leg["B"] / (leg["ms_per_step"] * 1e-3)) < 1e-3 * leg["samples_per_s"]
    assert len(json.dumps(j["small_shapes"])) < 1500


def test_cfg3_line_carries_the_panel_gemm_block():
    # the headline workload: the GRU input projections run as whole-N panels (csrc/gemm_panel.hip); the line prices the
    # kernel against the dense bf16 matrix peak, measured alone through the C-ABI op on operands of the workload's shape
    j = _run("--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--probe-rows", "2000000")
    gp = j["roofline_other"]["gru input projections (gemm_panel_kernel)"]
    assert gp["bound"] == "mfma" and gp["unit"] == "TFLOP/s" and gp["peak"] == 2500.0 and 0 < gp["frac"] <= 1.0
    assert abs(gp["frac"] - gp["achieved"] / gp["peak"]) < 1e-9 and abs(gp["achieved"] - 6 * gp["fp32_equivalent_tflops"]) < 1e-6 * gp["achieved"]
    assert gp["max_rel_err_vs_fp64"] < 5e-6 and "traffic" in gp and gp["avg_launch_ms"] > 0
    assert j["metric"] == "train samples/sec @ batch=1024" and j["config"]["workload"].startswith("cfg3")


def test_flags_change_what_they_say():
    j = _run("--steps", "6", "--warmup", "2", "--config", "cfg2", "--no-cpu-baseline", "--no-side", "--fresh-state")
    assert j["cpu_baseline"] is None and "value_best_case" not in j and j["config"]["live_row_frac"] < 1.0
    j = _run("--steps", "6", "--warmup", "2", "--config", "cfg2", "--no-cpu-baseline", "--no-side", "--graph")
    assert j["launch"].startswith("one captured hipGraph") and j["value"] > 0
    j = _run("--steps", "4", "--warmup", "1", "--config", "cfg2", "--no-cpu-baseline", "--force-sharded", "--global-batch", "256")
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and "value_best_case" not in j


def test_time_tiled_optimizer_line():
    """the headline's table optimizer (on by default from 256 MB of sweep traffic; forced here on the small shape) and
    the per-step sweep it replaces, timed in the same process"""
    j = _run("--steps", "20", "--warmup", "3", "--config", "cfg2", "--no-cpu-baseline", SCORE_ADAM_TILED_MIN_BYTES="0")
    assert j["config"]["table_optimizer"].startswith("time-tiled ApplyAdam, window 24")
    assert j["value_dense_adam_sweep"] > 0 and j["stages_ms"]["adam_catchup_batch_rows"] > 0
    assert any(k.startswith("adam_touched") for k in j["roofline_other"])
    j = _run("--steps", "6", "--warmup", "2", "--config", "cfg2", "--no-cpu-baseline", "--no-side", SCORE_ADAM_WINDOW="0")
    assert j["config"]["table_optimizer"].startswith("dense ApplyAdam sweep") and "adam_catchup_batch_rows" not in j["stages_ms"]


def test_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` (no launcher) starts N ranks itself, before anything touches the GPU.
    On this one-GPU box: refused with a clear message for N = 2 over RCCL; the same command line rehearsed with two gloo
    ranks on device 0 prints the one JSON line with n_gpus = 2."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    import torch
    if torch.cuda.device_count() < 2:
        assert out.returncode == 2 and "2 GPU" not in out.stdout and "GPU(s) visible" in out.stderr, out.stderr[-500:]
    j = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--config", "cfg2", SCORE_BENCH_DEVICE="0", SCORE_DIST_BACKEND="gloo")
    assert j["n_gpus"] == 2 and j["dist"]["backend"] == "gloo" and j["dist"]["world_size"] == 2 and j["rccl_ranks"] == 0
    # the all-to-all form was settled at set-up, and every rank reports its collectives
    assert "gloo" in j["dist"]["a2a_probe"]["form"] and len(j["dist"]["collectives_ms"]) == 2
    for t in j["dist"]["collectives_ms"]:
        assert {"a2a_row_requests", "a2a_rows", "a2a_row_grads", "all_reduce"} <= set(t) and t["a2a_rows"]["calls"] > 0
    assert j["cpu_baseline"] is None and j["ms_per_step_ranks"]["min"] <= j["ms_per_step_ranks"]["max"] == j["ms_per_step"]
    assert abs(j["value"] - 2 * 256 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


_RCCL_ONE_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from score_amd.synth import make_world
from score_amd.model import SCORE
from score_amd.dist import ShardedSCORE
world, kw = make_world("cfg2")
B = kw.pop("batch")
a, b = SCORE(seed=1111, **kw), ShardedSCORE(seed=1111, **kw)
assert torch.equal(a.table, b.backend.m.table) and torch.equal(a.w, b.backend.m.w)
bts = [world.batch(B, i) for i in range(6)]
la, lb = [], []
for i, bt in enumerate(bts):
    la.append(a.train(None, bt, 1e-3, 1e-4))
    lb.append(b.train(None, bt, 1e-3, 1e-4, next_batch=bts[i + 1] if i + 1 < len(bts) else None))
pa, _, _ = a.eval(None, bts[0], 1e-4)
pb, _, _ = b.eval(None, bts[0], 1e-4)
dp = max(abs(x - y) for x, y in zip(pa, pb))
dt = float((a.table - b.backend.m.table).abs().max().item())
print(json.dumps({"unsharded": la, "sharded": lb, "max_dpred": dp, "max_dtable": dt, "rccl_ranks": dist.get_world_size(),
                  "backend": dist.get_backend()}))
dist.destroy_process_group()
"""


def test_one_rank_through_rccl_equals_unsharded():
    """the row-sharded path with ONE rank over RCCL (backend nccl: all_to_all_single + all_reduce really go through the
    communicator) against the unsharded model: same seed, six different batches, dropout on (rank 0's seed stream is the
    single-device one).  The two differ only in the order row gradients are summed (owner-side accumulate vs pull
    scatter): losses to 1e-6 relative, predictions to 1e-5."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % {"root": ROOT, "port": port}], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["backend"] == "nccl" and j["rccl_ranks"] == 1
    for x, y in zip(j["unsharded"], j["sharded"]):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(x)), (j["unsharded"], j["sharded"])
    assert j["max_dpred"] < 1e-5 and j["max_dtable"] < 1e-5, j
    # and bench.py itself through that path says so in its line
    b = _run("--steps", "6", "--warmup", "2", "--config", "cfg2", "--no-cpu-baseline", "--force-sharded")
    assert b["rccl_ranks"] == 1 and b["dist"]["backend"] == "nccl" and b["n_gpus"] == 1
