# the kernel launch sequence of ONE steady-state training step (names, stream, duration), from a rocprofv3 kernel trace:
#   tools/kernel_sequence.sh <outdir-under-gpurun_out> <bench args...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --no-side --steps 40 --warmup 10 "$@" > $O/run.log 2>&1
find $O -name "*.db" -delete
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/tr/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one step = from one coattn_fwd_kernel to the next, taken near the end of the run
idx = [i for i, n in enumerate(names) if "ps_fwd_kernel" in n] or [i for i, n in enumerate(names) if "coattn_fwd_kernel" in n]
# (the middle of the TIMED region: the last dozen steps of a bench run carry the stage events of the stage table and are not
#  what the headline measures -- until round 5 this took idx[-3], a step without the look-ahead catch-up)
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
with open(sys.argv[1] + "/sequence.txt", "w") as o:
    o.write("%d launches between two forward kernels\n" % (b - a))
    for r in rows[a:b]:
        line = "%9.1f us  +%7.1f  stream %-2s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                              r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][-60:])
        o.write(line + "\n")
print(open(sys.argv[1] + "/sequence.txt").read())
PY
