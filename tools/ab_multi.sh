# Several environment arms on ONE box, interleaved: bash tools/ab_multi.sh <repeats> "<arm1 env>" "<arm2 env>" ... -- [bench args]
# an arm is a space-separated list of VAR=value (empty string = baseline); prints one line per run
N=$1; shift
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for arm in "${ARMS[@]}"; do
    env $arm python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[%s]' % '$arm', round(j['value']), round(j['ms_per_step'],4), 'p50', round(j.get('ms_per_step_p50',0),4), {k:round(v,3) for k,v in j['stages_ms'].items() if v})"
  done
done
