"""A variant build of libscore_hip.so for probes: python tools/build_variant.py <name> <source.hip,source.hip,...> <-Dflag ...>
recompiles the named sources with the extra flags (all others: the product build's objects) and links
score_amd/lib/libscore_hip_<name>.so; load it with SCORE_HIP_LIB=<that file>.  Never part of build()."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_amd import build as b      # noqa: E402


def build_variant(name, sources, flags):
    # variants keep results right (timing marks, alternative instruction choices); the stripped kernels of the *_probe.py
    # tools are single-kernel builds of their own and never link into a whole library
    assert not any("SCORE_PROBE_BUILD" in f for f in flags), "stripped kernels do not go into a loadable libscore_hip"
    b.build()
    objdir = os.path.join(b.HERE, "build", name)
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in b.SOURCES:
        if src in sources:
            op = os.path.join(objdir, src.replace(".hip", ".o"))
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + b.FLAGS + list(flags) + ["-c", os.path.join(b.CSRC, src), "-o", op]))
        else:
            op = os.path.join(b.HERE, "build", src.replace(".hip", ".o"))
        objs.append(op)
    assert all(p.wait() == 0 for p in procs)
    out = os.path.join(b.LIBDIR, "libscore_hip_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build_variant(sys.argv[1], sys.argv[2].split(","), sys.argv[3:]))
