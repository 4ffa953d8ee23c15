set -e
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_persample.py tests/test_gpu_adam_tiled.py tests/test_gpu_bad_ids.py tests/test_gpu_dist.py -x -q > $O/n_tests.log 2>&1 || { tail -40 $O/n_tests.log; exit 1; }
tail -3 $O/n_tests.log
python tools/host_calls.py tmall_default > $O/n_host_tmall.log 2>&1; cat $O/n_host_tmall.log
for c in tmall_default cfg2 taobao_default ccmr_default; do
timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 100 --no-cpu-baseline --no-side > $O/n_$c.json 2> $O/n_$c.err; python -c "
import json,sys; d=json.loads(open('$O/n_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
timeout -k 10 300 bash tools/kernel_sequence.sh r5/seq_tmall7 --config tmall_default > $O/seq_tmall7.log 2>&1; cat $O/seq_tmall7/sequence.txt
