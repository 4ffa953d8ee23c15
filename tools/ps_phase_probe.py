"""Where the per-sample whole-model kernels (csrc/persample.h) spend their time: runs a few steps on the -DPS_PHASE_TIMING
variant of the library (tools/build_variant.py phase ps_fwd.hip,ps_bwd.hip -DPS_PHASE_TIMING; one workgroup leaves a 100-MHz
timestamp at every phase boundary) and prints the phases' durations in us.
    SCORE_HIP_LIB=score_amd/lib/libscore_hip_phase.so python tools/ps_phase_probe.py [config]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from score_amd import _lib  # noqa: E402
from score_amd.model import SCORE  # noqa: E402
from score_amd.synth import make_world  # noqa: E402

FWD = ["gather+coattn+targets", "x-proj + q", "GRU (wave 0) | qz", "build [k, q*k]", "dense_3", "dense_4", "scores+softmax",
       "pooling", "bn1", "fc1", "fc2", "fc3+loss+dz2"]
BWD = ["loads", "dz1", "dbn", "dscore", "da2", "da1", "adzsum", "dinp + dqd", "d states / dq", "GRU (wave 0) | dquery", "dx",
       "coattn bwd", "dW slabs + targets"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tmall_default"
    world, kw = make_world(name)
    B = kw.pop("batch")
    if len(sys.argv) > 2:
        B = int(sys.argv[2])          # (e.g. 8: one workgroup per XCD -- nobody shares an L2 with anybody)
    m = SCORE(seed=1, **kw)
    bs = [m.device_batch(world.batch(B, i)) for i in range(4)]
    for i in range(12):
        m.train_async(bs[i % 4], 1e-3, 1e-4)
    torch.cuda.synchronize()
    lib = _lib.load()
    for which, names in (("fwd", FWD), ("bwd", BWD)):
        buf = (C.c_ulonglong * 32)()
        fn = getattr(lib, "score_ps_phase_read_" + which)
        fn.argtypes = [C.c_void_p]
        assert fn(buf) == 0
        ts = list(buf)[:len(names) + 1]
        print("%s: total %.1f us" % (which, (ts[-1] - ts[0]) / 100.0))
        for i, n in enumerate(names):
            print("   %-28s %6.1f us" % (n, (ts[i + 1] - ts[i]) / 100.0))


if __name__ == "__main__":
    main()
