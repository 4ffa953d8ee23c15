"""A/B build of libscore_hip with the correctly rounded sqrt / division in ApplyAdam (-DSCORE_ADAM_IEEE_DIV,
csrc/common.h score_adam1) -> score_amd/lib/libscore_hip_ieee.so.  Used with SCORE_HIP_LIB=<that file> to MEASURE what
v_sqrt_f32 / v_rcp_f32 cost in accuracy against the oracle's IEEE arithmetic (tests/test_gpu_adam_tiled.py, fixture g6;
numbers in profiles/r03_adam_oracle_pin.md).  Not part of build(): the product uses the fast form."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_amd import build as b      # noqa: E402

objdir = os.path.join(b.HERE, "build", "ieee")
os.makedirs(objdir, exist_ok=True)
objs, procs = [], []
for src in b.SOURCES:
    op = os.path.join(objdir, src.replace(".hip", ".o"))
    objs.append(op)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-DSCORE_ADAM_IEEE_DIV", "-c",
                                   os.path.join(b.CSRC, src), "-o", op]))
assert all(p.wait() == 0 for p in procs)
out = os.path.join(b.LIBDIR, "libscore_hip_ieee.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
