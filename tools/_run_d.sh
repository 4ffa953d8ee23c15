set -e
mkdir -p gpurun_out/r5
O=gpurun_out/r5
for v in "plan_ahead=True" "plan_ahead=False" "plan_ahead=True" "plan_ahead=False"; do
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side --set $v > $O/h_cfg3.json 2> $O/h_cfg3.err; python -c "
import json,sys; d=json.loads(open('$O/h_cfg3.json').read().strip().splitlines()[-1]); print('cfg3 $v', d['value'], d['ms_per_step'], sum(d['stages_ms'].values()))"
done
