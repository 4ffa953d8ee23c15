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


def test_heartbeat_gives_up_on_a_stalled_rank_and_names_where():
    """bench.py's per-rank heartbeat (several ranks: a hung collective must not eat the driver's timeout in silence): no
    progress for `stall` seconds -> a line naming rank, phase and the last collective entered, exit code 3"""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "class Cm: last = ('all_to_all[float32 x 7]', 41)\n"
            "hb = bench.Heartbeat(5, lambda: Cm(), every=0.5, stall=2.0)\n"
            "hb.tick('timed'); hb.tick()\n"
            "time.sleep(30)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-400:])
    assert "rank 5 made no progress" in out.stderr and "phase 'timed'" in out.stderr and "all_to_all[float32 x 7]" in out.stderr
    assert "heartbeat: rank 5 phase=timed" in out.stderr


def test_heartbeat_keeps_quiet_while_the_rank_progresses():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "hb = bench.Heartbeat(0, lambda: None, every=100.0, stall=1.5)\n"
            "for i in range(12):\n    hb.tick('timed'); time.sleep(0.3)\n"
            "hb.stop(); print('done')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip() == "done", (out.returncode, out.stderr[-400:])


def test_visible_gpus_counts_without_hip(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    # this container has no KFD driver: no device can be opened, whatever the environment says
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert bench.visible_gpus() == 0
