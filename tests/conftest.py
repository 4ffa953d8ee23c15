import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    """host cores this process may really use: the affinity mask AND the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        if q > 0:
            n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
    except Exception:
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is torch on the CPU.  A GPU box gives this process 16 cores of cgroup quota on a 256-thread host: by affinity
    # torch would start 256 OpenMP threads that the quota then throttles (round 6: the oracle-bound tests -- the 120-step
    # trajectory, the cfg-5 / full-size comparisons -- took a quarter of the GPU suite's wall clock that way)
    try:
        import torch
        torch.set_num_threads(min(_usable_cpus(), 16))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
