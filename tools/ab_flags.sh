# alternating A/B of two debug_flags values in one session: tools/ab_flags.sh <out-under-gpurun_out> <flagsA> <flagsB> <pairs> <bench args...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; A=$2; B=$3; N=$4; shift 4
mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  for f in $A $B; do
    python3 bench.py --no-cpu-baseline --no-side --debug-flags $f "$@" > $O/ab_${f}_$i.json 2>/dev/null
    python3 -c "
import json; d=json.loads(open('$O/ab_${f}_$i.json').read().strip().splitlines()[-1]); print('flags $f run $i: %.0f samples/s  %.4f ms  p50 %.4f' % (d['value'], d['ms_per_step'], d.get('ms_per_step_p50') or 0))"
  done
done
