# Several ARGUMENT arms of bench.py on ONE box, interleaved: bash tools/ab_args.sh <repeats> "<arm1 args>" "<arm2 args>" ... -- [common bench args]
# an arm may start with VAR=value words (environment) followed by bench.py arguments; prints one line per run
N=$1; shift
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for arm in "${ARMS[@]}"; do
    envs=""; rest=""
    for w in $arm; do case "$w" in [A-Z_]*=*) envs="$envs $w";; *) rest="$rest $w";; esac; done
    env $envs python3 bench.py --no-cpu-baseline --no-side "$@" $rest 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[%s]' % '$arm', round(j['value']), round(j['ms_per_step'],4), 'p50', round(j.get('ms_per_step_p50',0),4), {k:round(v,3) for k,v in j['stages_ms'].items() if v})"
  done
done
