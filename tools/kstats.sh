# per-kernel averages of one bench invocation: tools/kstats.sh <outdir-under-gpurun_out> <bench args...>
# KSTATS_TREE=<subdir>: run the bench.py of that source tree (tools/ab_trees.sh) instead of the repo's own
R=$GRAFT_REPO_ROOT/${KSTATS_TREE:-.}; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --no-cpu-baseline --no-side "$@" > $O/run.log 2>&1
find $O -name "*.db" -delete
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/st/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-64s n=%4s avg=%9.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep '^{' $O/run.log | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), {k:(round(v,3) if v else v) for k,v in j['stages_ms'].items()})"
