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
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # the headline's companions
    assert j["value_best_case"] > 0 and j["value_all_slices"] > 0 and 0 < j["best_case_live_row_frac"] <= 1
    lo = j["roofline_lowdup"]
    assert lo["bound"] == "hbm" and lo["algorithmic_bytes_per_launch"] > 0 and lo["avg_launch_ms"] > 0
    assert lo["expected_distinct_rows"] <= lo["row_uses"]
    ing = j["ingestion"]
    assert ing["device_assembly_samples_per_s"] > 0 and ing["nested_python_lists_samples_per_s"] > 0
    assert j["cpu_baseline_literal_tile"]["value"] > 0 and "materialised" in j["cpu_baseline_literal_tile"]["sample"]
    assert set(j["stages_ms"]) >= {"fwd_gather_coattn", "bwd_coattn_scatter", "adam_table_and_dense"}


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
