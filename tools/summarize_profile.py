#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small CSV/JSON summaries
committed under profiles/.

  python tools/summarize_profile.py stats  <rocprof_dir> <out.csv>          # --kernel-trace --stats
  python tools/summarize_profile.py pmc    <fetch_dir> <write_dir> <out.json> [tag]   # --pmc passes

PMC traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB, collected in separate passes; on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane
coalesced read stream, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (calibrated here on
adam_kernel: 4 read + 3 write streams of N*D*4 bytes reproduce to within 1 %).
"""
import collections
import csv
import glob
import json
import sys


def stats(src, out):
    import os
    f = max(glob.glob(src + "/*/*_kernel_stats.csv"), key=os.path.getmtime)      # (a re-used output directory keeps older processes' files)
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"]])


def _per_kernel(d, counter):
    import os
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
    return acc


def pmc(fetch_dir, write_dir, out, tag=""):
    f, w = _per_kernel(fetch_dir, "FETCH_SIZE"), _per_kernel(write_dir, "WRITE_SIZE")
    res = {"tag": tag, "note": "per-dispatch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                               "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)", "kernels": {}}
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fa = sum(f[k]) / len(f[k])
        wa = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
        res["kernels"][k] = {"dispatches": len(f[k]), "FETCH_SIZE_KiB": fa, "WRITE_SIZE_KiB": wa,
                             "hbm_bytes": (2 * fa + wa) * 1024}
    json.dump(res, open(out, "w"), indent=1)


def pmc_round(bench_fetch, bench_write, probe_fetch, probe_write, out, config="cfg3"):
    """The two-section file bench.py reads (committed_traffic): PMC bytes of the bench workload and of the
    low-duplication gather probe, stamped with the commit and the sha of embed.hip they were taken on."""
    import hashlib
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "score_amd", "csrc", "embed.hip"), "rb").read()).hexdigest()[:16]
    csrc = os.path.join(root, "score_amd", "csrc")
    shas = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16]
            for f in sorted(os.listdir(csrc)) if f.endswith(".hip") or f.endswith(".h")}
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
    except Exception:
        commit = None
    res = {}
    for key, fd, wd, wl in (("bench_workload", bench_fetch, bench_write,
                             "%s loader-shaped batches: bench.py --steps 6 --warmup 2 --no-side --no-cpu-baseline" % config),
                            ("gather_probe", probe_fetch, probe_write,
                             "low-duplication probe: bench.py --gather-probe-only (uniform ids over a 32 M-row table)")):
        tmp = tempfile.mktemp(suffix=".json")
        pmc(fd, wd, tmp, wl)
        sec = json.load(open(tmp))
        os.unlink(tmp)
        # steps the profiled run took: the fused gather runs exactly once per forward pass (round 4: bench.py also runs stage-table
        # and synchronous-train steps behind the timed region, so the count is read off the trace, not assumed)
        n_fwd = [v["dispatches"] for k, v in sec["kernels"].items() if "coattn_fwd_kernel" in k]
        sec.update(commit=commit, embed_hip_sha16=sha, source_sha16=shas, workload=wl,
                   steps_profiled=(n_fwd[0] if (key == "bench_workload" and n_fwd) else None))
        res[key] = sec
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc_round":
        pmc_round(*sys.argv[2:8])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
