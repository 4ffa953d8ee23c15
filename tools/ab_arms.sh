# several arms (each a set of environment assignments, "-" for none) alternating on ONE box:
#   bash tools/ab_arms.sh <repeats> "<arm1 env>" "<arm2 env>" ... -- [bench args]
N=$1; shift
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for arm in "${ARMS[@]}"; do
    if [ "$arm" = "-" ]; then E=""; else E="$arm"; fi
    env $E python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[$arm]', round(j['value']), round(j['ms_per_step'],4), {k:round(v,3) for k,v in j['stages_ms'].items() if v})"
  done
done
