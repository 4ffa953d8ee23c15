"""Build libscore_hip.so (hand-written HIP kernels + C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libscore_hip.so")
SOURCES = ["embed.hip", "gemm.hip", "gemm_bf16x3.hip", "gemm_panel.hip", "gru.hip", "gru_x3.hip", "gru_stream.hip", "head.hip", "head_fused.hip", "adam_tiled.hip", "scatter.hip", "sort.hip", "loader.hip", "ps_fwd.hip", "ps_bwd.hip", "step.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
         "-Wno-pass-failed", "-munsafe-fp-atomics"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the product library never carries a timing probe's stripped kernel (csrc/common.h: those switches need SCORE_PROBE_BUILD)
    if any("SCORE_PROBE_BUILD" in f for f in FLAGS) or "SCORE_PROBE_BUILD" in os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", ""):
        raise RuntimeError("score_amd.build: -DSCORE_PROBE_BUILD is for tools/*_probe.py only, never for libscore_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "score_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _stale(LIB, objs):
        # linked under a private name and renamed into place: another rank starting at the same moment (mp.spawn,
        # torchrun) either sees no file yet or a complete one, never a half-written .so
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    build_listpack(force)
    return LIB


def build_listpack(force=False):
    """The CPython helper that flattens nested feed lists (score_amd/cext/listpack.c): plain gcc, in-tree."""
    import sysconfig
    src = os.path.join(HERE, "cext", "listpack.c")
    out = os.path.join(LIBDIR, "_listpack.so")
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(out, [src]):
        tmp = "%s.%d.tmp" % (out, os.getpid())          # (atomic, as above: several ranks may get here together)
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-pthread",
                               "-I" + sysconfig.get_paths()["include"], src, "-o", tmp])
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
