# the launch stream's kernels of TWO consecutive steady-state steps with the idle gap in front of each, and for each gap the kernel
# (any stream) that ended last before the launch that followed it -- where does the chain wait, and for whom:
#   tools/kernel_timeline.sh <outdir-under-gpurun_out> <bench args...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --no-side --steps 40 --warmup 10 "$@" > $O/run.log 2>&1
find $O -name "*.db" -delete
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/tr/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
short = lambda r: r["Kernel_Name"].split("(")[0][-44:]
idx = [i for i, r in enumerate(rows) if "coattn_fwd_kernel" in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 2]
main = rows[a]["Stream_Id"]
t0 = S(rows[a])
out = []
periods = [(S(rows[idx[i + 1]]) - S(rows[idx[i]])) / 1e3 for i in range(len(idx) // 2 - 5, len(idx) // 2 + 5)]
out.append("periods (us) around the middle of the run: " + " ".join("%.0f" % p for p in periods))
prev = None
busy = 0.0
for r in rows[a:b]:
    if r["Stream_Id"] != main:
        continue
    gap = (S(r) - E(prev)) / 1e3 if prev is not None else 0.0
    # who ended last before this start, on another stream, within the gap
    last = None
    if prev is not None and gap > 3.0:
        cand = [q for q in rows if q["Stream_Id"] != main and E(prev) < E(q) <= S(r)]
        if cand:
            last = max(cand, key=E)
    busy += (E(r) - S(r)) / 1e3
    out.append("%9.1f  gap %6.1f  run %6.1f  %s%s" % ((S(r) - t0) / 1e3, gap, (E(r) - S(r)) / 1e3, short(r),
               ("   <- %s (stream %s) ended %.1f us before" % (short(last), last["Stream_Id"], (S(r) - E(last)) / 1e3)) if last else ""))
    prev = r
out.append("launch-stream kernel time over the two steps: %.1f us" % busy)
open(sys.argv[1] + "/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
