"""`python bench.py --gpus N` without a launcher must decide BEFORE touching the GPU: with fewer
than N devices visible it refuses with exit code 2 and a message; it never raises from inside a half-started run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_refused_without_devices():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert "--gpus 4 asked for, 0 GPU(s) visible" in out.stderr and out.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
