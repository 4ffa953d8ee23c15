set -e
mkdir -p gpurun_out/r5
O=gpurun_out/r5
python tools/host_calls.py tmall_default > $O/d_host_tmall.log 2>&1; cat $O/d_host_tmall.log
timeout -k 10 300 bash tools/kernel_sequence.sh r5/seq_tmall2 --config tmall_default > $O/seq_tmall2.log 2>&1; cat $O/seq_tmall2/sequence.txt
for c in tmall_default ccmr_default; do
timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 100 --no-cpu-baseline --no-side > $O/d_$c.json 2> $O/d_$c.err; python -c "
import json,sys; d=json.loads(open('$O/d_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
