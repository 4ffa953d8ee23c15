set -e
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_persample.py tests/test_gpu_adam_tiled.py tests/test_gpu_bad_ids.py tests/test_gpu_trajectory.py tests/test_gpu_harness.py tests/test_gpu_bench_contract.py -x -q > $O/l_tests.log 2>&1 || { tail -40 $O/l_tests.log; exit 1; }
tail -3 $O/l_tests.log
python tools/host_calls.py tmall_default > $O/l_host_tmall.log 2>&1; cat $O/l_host_tmall.log
for c in tmall_default cfg2 taobao_default ccmr_default; do
timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 100 --no-cpu-baseline --no-side > $O/l_$c.json 2> $O/l_$c.err; python -c "
import json,sys; d=json.loads(open('$O/l_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
timeout -k 10 200 python bench.py --config tmall_default --steps 2000 --warmup 100 --no-cpu-baseline --no-side --set fast_step=False > $O/l_tm_off.json 2> $O/l_tm_off.err; python -c "
import json,sys; d=json.loads(open('$O/l_tm_off.json').read().strip().splitlines()[-1]); print('tmall fast_step off', d['value'], d['ms_per_step'])"
