# A/B of an environment switch on ONE box, alternating runs (box-to-box spread is ~1 %, as large as most single changes):
#   bash tools/ab_env.sh "<VAR=value>" <repeats> [bench args...]      prints ms/step and the stage table of every run
V=$1; N=$2; shift; shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for mode in base switch; do
    if [ $mode = switch ]; then export $V; else unset ${V%%=*}; fi
    python3 bench.py --no-cpu-baseline --no-side "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$mode', round(j['value']), round(j['ms_per_step'],4), {k:round(v,3) for k,v in j['stages_ms'].items() if v})"
  done
done
