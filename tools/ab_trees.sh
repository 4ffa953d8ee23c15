# Several source TREES on ONE box, interleaved (A/B of a change that spans host code and library):
#   bash tools/ab_trees.sh <repeats> <tree A (relative to the repo root, "." = the repo itself)> <tree B> [more trees] -- [bench args]
N=$1; shift
TREES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do TREES+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq 1 $N); do
  for t in "${TREES[@]}"; do
    ( cd $GRAFT_REPO_ROOT/$t && python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[%s]' % '$t', round(j['value']), round(j['ms_per_step'],4), 'p50', round(j.get('ms_per_step_p50',0),4), {k:round(v,3) for k,v in j['stages_ms'].items() if v})" )
  done
done
