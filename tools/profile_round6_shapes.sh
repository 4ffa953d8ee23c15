# Round-6 bench lines of every shape but the headline (run on the GPU box through gpurun; outputs under gpurun_out/$1)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r06x}
mkdir -p $O
cd $R
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2: %.0f samples/s  %.4f ms  p50 %.4f' % (d['value'], d['ms_per_step'], d.get('ms_per_step_p50') or 0))"; }
for c in cfg2 tmall_default taobao_default ccmr_default; do
  python3 bench.py --no-cpu-baseline --no-side --config $c --steps 2000 --warmup 200 > $O/bench_$c.json 2>>$O/err.log && show $O/bench_$c.json $c
done
python3 bench.py --no-cpu-baseline --no-side --config cfg5_taobao --steps 10 --warmup 3 --batches 2 > $O/bench_cfg5_taobao.json 2>>$O/err.log && show $O/bench_cfg5_taobao.json cfg5_taobao
python3 bench.py --no-cpu-baseline --no-side --config cfg5_tmall --steps 6 --warmup 2 --batches 2 > $O/bench_cfg5_tmall.json 2>>$O/err.log && show $O/bench_cfg5_tmall.json cfg5_tmall
for b in 128 256 512; do
  python3 bench.py --force-sharded --global-batch $b --no-cpu-baseline --no-side --steps 300 --warmup 30 > $O/fs_b$b.json 2>>$O/err.log && show $O/fs_b$b.json cfg4rank_b$b
done
python3 bench.py --force-sharded --global-batch 512 --config cfg5_tmall --no-cpu-baseline --no-side --steps 10 --warmup 3 --batches 2 > $O/cfg5_fs_b512.json 2>>$O/err.log && show $O/cfg5_fs_b512.json cfg5_tmall_rank_b512
python3 bench.py --force-sharded --no-cpu-baseline --no-side --steps 100 --warmup 10 > $O/fs_b1024.json 2>>$O/err.log && show $O/fs_b1024.json cfg3_sharded_1rank
